"""Golden fixture: the reference CLI's OBSERVABLE side effects when the input directory holds a bad entry.

Runs the real ``/root/reference/infer_alns.py`` (build container only; empty ``dendropy`` stub as in
``gen_golden.py``) on small directories and records, for each scenario, the order ``glob`` listed the directory
in (the reference processes entries in that order, infer_alns.py:97), which output files exist afterwards, the
exception type and the last line of the traceback:

* ``bad_extension``: a ``.txt`` entry among FASTA files -> ``ValueError`` at that entry (infer_alns.py:100-103),
  every entry listed before it has its ``.phy``;
* ``bad_residue``: a FASTA file with a byte outside the alphabet -> ``KeyError`` from ``load_alignment``
  (data.py:26) at that file, files before it written;
* ``too_many_seqs``: an alignment of 201 sequences -> ``ValueError`` from ``adaptable_seq2pair`` (model.py:24-28) inside
  the forward of that file, files before it written;
* ``single_sequence``: an alignment of ONE sequence parses (``fasta_edge.json``) and then fails inside the forward with a
  ``RuntimeError`` - there are no pairs, and ``attention.py:193`` cannot view the empty tensor - files before it written.

Scenarios already in the fixture are kept as they are (the listing order is the file system's); delete the file to
regenerate all of them.

Output: ``tests/golden/cli_bad_entry.json`` (data only).  ``tests/test_scheduler.py`` and
``tests/test_cli_gpu.py`` hold this build's CLI to the same rule: outputs == the entries in front of the offender,
in the order the CLI was given them.
    python oracle/gen_golden_cli_errors.py
"""
import json
import os
import subprocess
import sys
import tempfile
from glob import glob

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)


def main():
    import numpy as np
    from phyloformer_amd.msa_sim import simulate_batch, to_fasta
    stub = tempfile.mkdtemp(prefix="pf_stub_")
    os.makedirs(os.path.join(stub, "dendropy"))
    open(os.path.join(stub, "dendropy", "__init__.py"), "w").close()
    env = dict(os.environ, PYTHONPATH=f"{stub}:{REF}")
    alns = simulate_batch(6, 5, 12, seed=77)
    names = ["c.fa", "a.fasta", "e.FA", "b.fa", "f.fa", "d.fa"]
    path = os.path.join(REPO, "tests", "golden", "cli_bad_entry.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    # the listing order is the file system's (glob does not sort): try offender names until one lands strictly
    # inside the listing, so that the fixture shows files on both sides of it
    big = "".join(f">t{k}\n{'ARN' if k % 2 else 'ARD'}\n" for k in range(201)).encode()      # 201 sequences: over SEQ2PAIR's 200
    todo = [("bad_extension", f"notes{k}.txt", b"not an alignment\n") for k in range(12)] + \
           [("bad_residue", f"bad{k}.fa", b">s0\nARNDB\n>s1\nARNDC\n") for k in range(12)] + \
           [("too_many_seqs", f"big{k}.fa", big) for k in range(12)] + \
           [("single_sequence", f"one{k}.fa", b">s0\nARNDCQEGHILK\n") for k in range(12)]
    for scenario, offender, content in todo:
        if scenario in out:
            continue
        with tempfile.TemporaryDirectory() as td:
            ind, outd = os.path.join(td, "in"), os.path.join(td, "out")
            os.makedirs(ind)
            for k, name in enumerate(names[:3]):
                open(os.path.join(ind, name), "w").write(to_fasta(alns[k]))
            open(os.path.join(ind, offender), "wb").write(content)
            for k, name in enumerate(names[3:]):
                open(os.path.join(ind, name), "w").write(to_fasta(alns[3 + k]))
            order = [os.path.basename(p) for p in glob(f"{ind}/*")]
            if not 1 <= order.index(offender) <= len(order) - 2:
                continue
            res = subprocess.run([sys.executable, os.path.join(REF, "infer_alns.py"), "-o", outd,
                                  os.path.join(REF, "models/pf_base.ckpt"), ind], env=env, cwd=td,
                                 capture_output=True, text=True)
            last = [l for l in res.stderr.strip().splitlines() if l.strip()][-1]
            written = sorted(os.listdir(outd))
            before = [os.path.splitext(n)[0] + ".phy" for n in order[:order.index(offender)]]
            assert sorted(before) == written, (order, written)          # the rule the build mirrors
            out[scenario] = {"listing_order": order, "offender": offender, "returncode": res.returncode,
                             "outputs": written, "exception": last.split(":")[0],
                             "last_line": last.replace(ind, "<in>")}
            print(scenario, order, "->", written, "|", last.replace(ind, "<in>"))
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
