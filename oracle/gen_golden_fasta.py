#!/usr/bin/env python3
"""Reference-generated fixtures for the FASTA boundary (``load_alignment``, phyloformer/data.py:11-31).

Build container only.  Every case is a small FASTA byte string written by this script (test input, not
reference text); the REFERENCE's ``load_alignment`` is run on it and what it returns - the ids and the
residue indices recovered from its one-hot tensor - or the class of the exception it raises is recorded
in ``tests/golden/fasta_edge.json``.  ``tests/test_host.py`` replays the cases through the Python mirror
and the native parser.

    python oracle/gen_golden_fasta.py
"""
import base64
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import _import_reference, GOLD  # noqa: E402

CASES = {
    "plain": b">a\nARND\n>b\nCQEG\n",
    "multi_line_records": b">s1\nARN\nDCQ\n>s2\nEGH\nILK\n",
    "crlf": b">a\r\nARND\r\n>b\r\nCQEG\r\n",
    "padded_ids_and_blank_lines": b">  tip 1  \nARND\n\n>\ttip_2\t\n\nCQEG\n   \n",
    "unknown_and_gap": b">a\nAX-R\n>b\n--XX\n",
    "no_trailing_newline": b">a\nAR\n>b\nND",
    "single_sequence": b">only\nARNDCQEG\n",
    "empty_header": b">\nAR\n>\nND\n",
    "header_only_records": b">a\n>b\n",
    "lowercase_residue": b">a\nArnd\n>b\nCQEG\n",
    "illegal_byte_B": b">a\nABND\n>b\nCQEG\n",
    "digit_in_sequence": b">a\nAR1D\n>b\nCQEG\n",
    "ragged": b">a\nARND\n>b\nCQ\n",
    "data_before_header": b"ARND\n>a\nCQEG\n",
    "empty_file": b"",
    "only_blank_lines": b"\n\n  \n",
    "inner_whitespace_in_sequence": b">a\nAR ND\n>b\nCQEGH\n",
    "gt_inside_line": b">a\nAR>D\n>b\nCQEG\n",
    "all_alphabet": b">a\nARNDCQEGHILKMFPSTWYVX-\n>b\n-XVYWTSPFMKLIHGEQCDNRA\n",
    "utf8_id": ">täxon\nAR\n>b\nND\n".encode("utf8"),
}


def main():
    torch, _Phyloformer, load_alignment, _stub = _import_reference()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in CASES.items():
            path = os.path.join(tmp, name + ".fa")
            with open(path, "wb") as fh:
                fh.write(data)
            rec = {"fasta_b64": base64.b64encode(data).decode()}
            try:
                x, ids = load_alignment(path)
                # x: int64 [22, L, N] one-hot (data.py:28-29) -> indices [N, L]
                rec["ids"] = list(ids)
                rec["shape"] = list(x.shape)
                rec["indices"] = x.argmax(0).T.tolist() if x.numel() else []
            except Exception as exc:  # noqa: BLE001 - the class is the fixture
                rec["raises"] = type(exc).__name__
            out[name] = rec
            print(name, "->", rec.get("raises") or (rec["shape"], rec["ids"]))
    with open(os.path.join(GOLD, "fasta_edge.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
