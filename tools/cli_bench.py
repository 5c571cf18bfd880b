"""End-to-end CLI throughput: write K synthetic FASTA files, run infer_alns.py on the directory.
    python tools/cli_bench.py [--n 256] [--seqs 60] [--sites 500] [--trees] [--extra "--batch 1 --python-io"]"""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phyloformer_amd.fasta import ALPHABET
from phyloformer_amd.msa_sim import simulate_batch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--seqs", type=int, default=60)
ap.add_argument("--sites", type=int, default=500)
ap.add_argument("--extra", default="")
ap.add_argument("--trees", action="store_true", help="pass -t: <stem>.nj.nwk beside every <stem>.phy")
a = ap.parse_args()
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (tmpfs when there is one: the box's overlay file system throttles after the first few thousand small files - a second
# run in the same box measured 3,100 alignments/s with or without trees where the first made 10,500; profiles/r06r_*)
tmp = tempfile.mkdtemp(prefix="pfcli_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and not os.environ.get("PF_CLI_BENCH_TMP") else os.environ.get("PF_CLI_BENCH_TMP"))
ind, outd = os.path.join(tmp, "in"), os.path.join(tmp, "out")
os.makedirs(ind)
base = simulate_batch(8, a.seqs, a.sites, seed=3)
lut = np.frombuffer(ALPHABET, dtype=np.uint8)
for k in range(a.n):
    idx = np.roll(base[k % 8], k // 8, axis=1)
    with open(os.path.join(ind, f"aln{k:05d}.fa"), "wb") as fh:
        for i, row in enumerate(idx):
            fh.write(b">taxon_%d\n" % i + lut[row].tobytes() + b"\n")
t0 = time.perf_counter()
r = subprocess.run([sys.executable, os.path.join(repo, "infer_alns.py"), os.path.join(repo, "models/pf.ckpt"),
                    ind, "-o", outd, "--bench", *(["-t"] if a.trees else []), *a.extra.split()], capture_output=True, text=True)
wall = time.perf_counter() - t0
rep = [l for l in r.stderr.splitlines() if l.startswith("{")]
print(json.dumps({"files": a.n, "shape": [a.seqs, a.sites], "trees": a.trees, "extra": a.extra, "process_wall_s": round(wall, 3),
                  "returncode": r.returncode, "report": json.loads(rep[-1]) if rep else r.stderr[-500:]}))
assert len(os.listdir(outd)) == a.n * (2 if a.trees else 1)
shutil.rmtree(tmp)
