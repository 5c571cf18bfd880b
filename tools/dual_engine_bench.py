"""Do two engines on two host threads (two streams) hide the host-side gaps of the synchronous pf_forward?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phyloformer_amd.engine import Engine
from phyloformer_amd.weights import load_weights
from phyloformer_amd.msa_sim import simulate_batch


def main():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w = load_weights(os.path.join(repo, "models/pf.ckpt"))
    idx = np.ascontiguousarray(np.resize(simulate_batch(8, 60, 500, seed=3), (16, 60, 500)))
    for nthreads in (1, 2, 3):
        engs = [Engine(w) for _ in range(nthreads)]
        for e in engs: e.forward(idx)
        steps = 12
        def work(e):
            for _ in range(steps): e.forward(idx)
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(e,)) for e in engs]
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        print(f"{nthreads} engine thread(s): {nthreads * steps * 16 / dt:7.1f} aln/s")
        for e in engs: e.close()


if __name__ == "__main__":
    main()
