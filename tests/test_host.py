"""Host-side mirror of the reference's Python boundary: FASTA, PHYLIP, .ckpt, blob, NJ, sharding."""
import os
import re

import numpy as np
import pytest

from phyloformer_amd import ckpt, fasta, phylip, weights as W
from phyloformer_amd.dist import alignment_range, site_range, site_ranges
from phyloformer_amd.nj import neighbor_joining

CKPTS = ["pf", "pf_base", "pf_indel", "pf_cherry", "pf_selreg"]


# ---- FASTA (reference: phyloformer/data.py:11-31) --------------------------------------------
def test_fasta_reference_msa(repo):
    idx, ids = fasta.load_alignment(os.path.join(repo, "data/testdata/msas/0_20_tips.fa"))
    assert idx.shape == (20, 250) and idx.dtype == np.uint8
    assert ids[0] == "T4" and len(ids) == 20          # trailing blanks of the header are stripped
    assert idx.max() < 20                             # test MSAs have no X / gap


def test_fasta_rules():
    idx, ids = fasta.parse_fasta(b">a \nAR\nND\n\n> b\nX-\n  CQ  \n")
    assert ids == ["a", " b"]                         # line.strip() then line[1:], data.py:21-22
    assert idx.tolist() == [[0, 1, 2, 3], [20, 21, 4, 5]]
    with pytest.raises(KeyError):                     # LOOKUP[char], data.py:26
        fasta.parse_fasta(b">a\nAZB\n")
    with pytest.raises(KeyError):
        fasta.parse_fasta(b">a\nar\n")                # lower case is not in the alphabet
    with pytest.raises(ValueError):                   # ragged → torch.tensor fails, data.py:28
        fasta.parse_fasta(b">a\nAR\n>b\nA\n")
    with pytest.raises(IndexError):
        fasta.parse_fasta(b"AR\n>a\nAR\n")


def test_one_hot_round_trip():
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 22, size=(5, 9)).astype(np.uint8)
    oh = fasta.one_hot(idx)
    assert oh.shape == (22, 9, 5) and oh.dtype == np.int64 and (oh.sum(0) == 1).all()
    assert np.array_equal(fasta.from_one_hot(oh[None].astype(np.float32))[0], idx)


# ---- PHYLIP (reference: infer_alns.py:14-25) --------------------------------------------------
def test_phylip_text_matches_reference_cli(repo, golden):
    g = golden("e2e_testdata.npz")
    _idx, ids = fasta.load_alignment(os.path.join(repo, "data/testdata/msas/0_20_tips.fa"))
    dm, text = phylip.vec_to_phylip(g["pf_base/0_20_tips"], ids)
    with open(os.path.join(repo, "tests/golden/0_20_tips.pf_base.phy")) as fh:
        assert text == fh.read()                      # byte-identical to the reference CLI's file
    assert dm.shape == (20, 20) and np.array_equal(dm, dm.T) and (np.diag(dm) == 0).all()
    assert text.splitlines()[1].startswith("T4 0.0000000000 0.3387762010")


def test_phylip_pair_order():
    d = np.arange(1, 7, dtype=np.float32)
    dm = phylip.vec_to_matrix(d, 4)
    assert dm[0].tolist() == [0, 1, 2, 3] and dm[1, 2:].tolist() == [4, 5] and dm[2, 3] == 6
    assert phylip.vec_to_matrix(np.float32(2.5), 2).tolist() == [[0, 2.5], [2.5, 0]]


# ---- checkpoints (reference: infer_alns.py:71-82) ----------------------------------------------
@pytest.mark.parametrize("name", CKPTS)
def test_ckpt_reader_matches_torch_load(repo, name):
    torch = pytest.importorskip("torch")
    path = os.path.join(repo, "models", f"{name}.ckpt")
    sd, hp = ckpt.load_state_dict(path)
    ref = torch.load(path, map_location="cpu", weights_only=True)
    assert hp == ref["hyper_parameters"] == {"nb_blocks": 6, "nb_heads": 4, "embed_dim": 64, "dropout": 0.0}
    want = {k.replace("model.", ""): v for k, v in ref["state_dict"].items() if k != "model.seq2pair"}
    assert set(sd) == set(want) and len(sd) == 160
    for k, v in want.items():
        assert np.array_equal(sd[k], v.numpy()), k
    w = W.from_state_dict(sd)
    assert (w.n_blocks, w.n_heads, w.embed_dim, w.n_params) == (6, 4, 64, 308449)


def test_ckpt_refuses_code_and_garbage(tmp_path):
    import pickle
    import zipfile
    p = tmp_path / "evil.ckpt"
    with zipfile.ZipFile(p, "w") as z:
        z.writestr("evil/data.pkl", pickle.dumps({"state_dict": {"a": os.system}}, protocol=2))
        z.writestr("evil/byteorder", "little")
    sd, _ = ckpt.load_state_dict(p)       # os.system decodes to an inert placeholder, never called
    assert sd == {}
    q = tmp_path / "garbage.ckpt"
    q.write_bytes(b"not a zip")
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_ckpt(q)


def test_weight_blob_layout(weights):
    w = weights("pf")
    blob = w.blob()
    assert blob.dtype == np.float32 and blob.size == 308449
    assert np.array_equal(blob[:64 * 22].reshape(64, 22), w["embedding_block.0.weight"])
    assert blob[-1] == w["pwFNN.0.bias"][0]
    bad = dict(w.tensors)
    bad["attention_blocks.0.ffn.0.weight"] = np.zeros((128, 64), np.float32)
    with pytest.raises(ckpt.CheckpointError):
        W.from_state_dict(bad)


# ---- C ABI: the library builds, loads and exports every declared symbol -------------------------
def test_abi_symbols_exported(repo):
    from phyloformer_amd import build, engine
    build.build()
    lib = engine.load_library()
    header = open(os.path.join(repo, "include/phyloformer_amd.h")).read()
    declared = set(re.findall(r"^(?:int64_t|uint64_t|int32_t|int|void|const char\*)\s+(pf_\w+)\(", header, re.M))
    assert declared and declared == set(engine.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    abi = int(re.search(r"#define PF_ABI_VERSION (\d+)", header).group(1))
    assert lib.pf_abi_version() == abi == engine.ABI_VERSION
    assert int(re.search(r"#define PF_UNIQUE_ID_BYTES (\d+)", header).group(1)) == engine.UNIQUE_ID_BYTES
    assert lib.pf_blob_len(6, 4, 64) == 308449


def test_every_option_the_library_accepts_is_documented_in_the_header(repo):
    """include/phyloformer_amd.h lists the options of pf_set_option; csrc/pf_lib.hip implements them: the same set."""
    import re
    h = open(os.path.join(repo, "include", "phyloformer_amd.h")).read()
    blk = h[h.index("/* Options (before or between forwards):"):h.index("int pf_set_option")]
    documented = set(re.findall(r'^ \*   "(\w+)"', blk, re.M))
    src = open(os.path.join(repo, "phyloformer_amd", "csrc", "pf_lib.hip")).read()
    fn = src[src.index("int pf_set_option("):]
    implemented = set(re.findall(r'k == "(\w+)"', fn[:fn.index("\n}\n")]))
    assert documented == implemented, (documented - implemented, implemented - documented)


def test_no_cpu_fallback_without_gpu(weights):
    """On a machine without a gfx950 device pf_create must fail loudly."""
    import ctypes
    from phyloformer_amd import engine
    try:
        ndev = ctypes.CDLL("libamdhip64.so").hipGetDeviceCount
        n = ctypes.c_int(0)
        has_gpu = ndev(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(engine.EngineError, match="no HIP device|no CPU fallback"):
        engine.Engine(weights("pf"))


# ---- sharding helpers / NJ ---------------------------------------------------------------------
def test_site_and_alignment_ranges():
    assert site_ranges(2000, 8) == [(250 * r, 250 * (r + 1)) for r in range(8)]
    r = site_ranges(500, 8)
    assert r[0] == (0, 63) and r[-1] == (441, 500) and sum(b - a for a, b in r) == 500
    assert site_ranges(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert [alignment_range(10, 4, k) for k in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert site_range(7, 1, 0) == (0, 7)


def test_neighbor_joining_recovers_additive_tree():
    # ((a:1,b:2):1,(c:3,d:4):1): additive distances → NJ must return exactly these branch lengths
    ids = ["a", "b", "c", "d"]
    d = np.array([[0, 3, 6, 7], [3, 0, 7, 8], [6, 7, 0, 7], [7, 8, 7, 0]], float)
    nwk = neighbor_joining(d, ids)
    assert nwk.endswith(";\n") and all(t in nwk for t in ids)
    lens = sorted(float(x) for x in re.findall(r":([0-9.eE+-]+)", nwk))
    assert lens == [1.0, 2.0, 2.0, 3.0, 4.0]


# ---- native file-format helpers (csrc/pf_hostio.cpp) vs the Python mirrors of the reference ----------

def _random_fasta(rng, n, l, wrap=None, crlf=False, pad_ids=False, blank_lines=False):
    from phyloformer_amd.fasta import ALPHABET
    out = []
    for i in range(n):
        seq = bytes(ALPHABET[c] for c in rng.integers(0, 22, l))
        name = b"seq_%d" % i + (b"   " if pad_ids else b"")
        out.append(b">" + name)
        if wrap:
            out += [seq[k:k + wrap] for k in range(0, l, wrap)]
        else:
            out.append(seq)
        if blank_lines:
            out.append(b"  \t ")
    eol = b"\r\n" if crlf else b"\n"
    return eol.join(out) + (eol if n % 2 else b"")


def test_native_fasta_matches_python_parser(repo):
    MSA_DIR = os.path.join(repo, "data/testdata/msas")
    from phyloformer_amd import fasta, hostio
    rng = np.random.default_rng(5)
    for n, l, kw in [(2, 1, {}), (5, 33, dict(wrap=10)), (20, 250, dict(pad_ids=True)),
                     (7, 64, dict(crlf=True, wrap=60)), (3, 17, dict(blank_lines=True)), (60, 500, {})]:
        data = _random_fasta(rng, n, l, **kw)
        a, ids_a = fasta.parse_fasta(data)
        b, ids_b = hostio.parse_fasta(data)
        assert ids_a == ids_b
        assert a.shape == b.shape == (n, l) and a.dtype == b.dtype == np.uint8
        assert np.array_equal(a, b)
    # the reference's test alignments
    for name in sorted(os.listdir(MSA_DIR))[:4]:
        a, ids_a = fasta.load_alignment(os.path.join(MSA_DIR, name))
        b, ids_b = hostio.load_alignment(os.path.join(MSA_DIR, name))
        assert ids_a == ids_b and np.array_equal(a, b)


@pytest.mark.parametrize("data,exc", [
    (b">a\nARND\n>b\nARNB\n", KeyError),          # byte outside the alphabet (data.py:26)
    (b">a\nARND\n>b\narnd\n", KeyError),          # lower case is not in the alphabet either
    (b">a\nARND\n>b\nARN\n", ValueError),         # ragged
    (b"ARND\n>a\nARND\n", IndexError),            # residues before the first header
    (b"\n\n", RuntimeError),                      # nothing: one_hot refuses the empty tensor (data.py:28)
    (b">a\n>b\n", RuntimeError),                  # records without residues: same
])
def test_native_fasta_errors_match_python_parser(data, exc):
    from phyloformer_amd import fasta, hostio
    with pytest.raises(exc) as e1:
        fasta.parse_fasta(data)
    with pytest.raises(exc) as e2:
        hostio.parse_fasta(data)
    if exc is KeyError:
        assert e1.value.args == e2.value.args


def test_fasta_edge_cases_match_reference_fixtures(repo):
    """tests/golden/fasta_edge.json holds what the REFERENCE's load_alignment (phyloformer/data.py:11-31)
    returns or raises on 20 edge-case files (generator: oracle/gen_golden_fasta.py); the Python mirror
    and the native parser must do the same - same ids, same indices, same exception class."""
    import base64
    import json
    from phyloformer_amd import fasta, hostio
    with open(os.path.join(repo, "tests", "golden", "fasta_edge.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 20
    excs = {"KeyError": KeyError, "ValueError": ValueError, "IndexError": IndexError, "RuntimeError": RuntimeError}
    for name, rec in cases.items():
        data = base64.b64decode(rec["fasta_b64"])
        for parse in (fasta.parse_fasta, hostio.parse_fasta):
            if "raises" in rec:
                with pytest.raises(excs[rec["raises"]]):
                    parse(data)
                    pytest.fail(f"{name}: {parse.__module__} accepted what the reference rejects")
            else:
                idx, ids = parse(data)
                assert ids == rec["ids"], (name, parse.__module__)
                assert idx.dtype == np.uint8 and idx.tolist() == rec["indices"], (name, parse.__module__)
                assert list(fasta.one_hot(idx).shape) == rec["shape"]


def test_native_phylip_is_byte_identical_to_python_writer():
    from phyloformer_amd import hostio
    from phyloformer_amd.phylip import vec_to_phylip
    rng = np.random.default_rng(11)
    for n in (2, 3, 20, 61):
        p = n * (n - 1) // 2
        preds = (rng.random(p) * rng.choice([1e-6, 1.0, 30.0], p)).astype(np.float32)
        preds[0] = 0.0
        ids = [f"t{i}_é" if i % 7 == 0 else f"taxon{i}" for i in range(n)]
        _dm, text = vec_to_phylip(preds, ids)
        assert hostio.format_phylip(preds, ids) == text.encode("utf8")
    with pytest.raises(ValueError):
        hostio.format_phylip(np.zeros(4, np.float32), ["a", "b", "c"])


def test_native_fasta_parser_agrees_with_python_on_arbitrary_bytes():
    """Property test: for any byte string over a FASTA-like alphabet the native parser returns what the
    Python mirror of data.py:11-31 returns, or raises the same exception class."""
    from hypothesis import given, settings, strategies as st
    from phyloformer_amd import hostio

    alphabet = st.sampled_from([b">", b"\n", b"\r\n", b" ", b"\t", b"A", b"R", b"-", b"X", b"V", b"a", b"B", b"seq", b"\x0b"])

    @settings(max_examples=400, deadline=None)
    @given(st.lists(alphabet, max_size=40).map(b"".join))
    def check(data):
        def run(fn):
            try:
                idx, ids = fn(data)
                return ("ok", idx.shape, idx.tobytes(), tuple(ids))
            except (KeyError, ValueError, IndexError, RuntimeError) as exc:
                return (type(exc).__name__, exc.args if isinstance(exc, KeyError) else None)
        assert run(hostio.parse_fasta) == run(fasta.parse_fasta)

    check()


def test_header_is_plain_c_and_links_from_c(repo, tmp_path):
    """include/phyloformer_amd.h must be consumable by a C compiler (it is the drop-in boundary), and a C
    program must link against the shared library and call the handle-free entry points."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    from phyloformer_amd.engine import LIB_PATH
    src = tmp_path / "abi.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "phyloformer_amd.h"
int main(void) {
    if (pf_abi_version() != PF_ABI_VERSION) return 1;
    if (pf_blob_len(6, 4, 64) != 308449ull) return 2;          /* parameter count of the shipped checkpoints */
    const char* fa = ">a\nARND\n>b\nAR-X\n";
    uint8_t idx[8]; int64_t spans[4]; int32_t n = 0, l = 0; int64_t detail = 0;
    if (pf_parse_fasta(fa, (int64_t)strlen(fa), idx, 8, spans, 2, &n, &l, &detail) != PF_OK) return 3;
    if (n != 2 || l != 4 || idx[0] != 0 || idx[6] != 21 || idx[7] != 20) return 4;
    const float d[1] = {0.25f}; const char* ids[2] = {"a", "b"}; char out[128];
    int64_t w = pf_format_phylip(d, 2, ids, out, sizeof out);
    if (w <= 0 || strncmp(out, "2\na 0.0000000000 0.2500000000\nb 0.2500000000 0.0000000000\n", (size_t)w) != 0) return 5;
    printf("abi ok\n");
    return 0;
}
""")
    exe = tmp_path / "abi"
    libdir = os.path.dirname(LIB_PATH)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(repo, "include"), str(src),
                        "-o", str(exe), "-L", libdir, "-l:libphyloformer_amd.so", f"-Wl,-rpath,{libdir}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "abi ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
