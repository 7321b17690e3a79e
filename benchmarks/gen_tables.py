#!/usr/bin/env python3
"""Generator-table precompute tool: evaluates the reference's CAFs `h`, `take n gs`, `take n hs`
(/root/reference/Bulletproofs/Utils.hs:14-25,38-47) once on the GPU and writes them to a file that later processes
install with bp_generators_load / Engine.load_generators.

  python benchmarks/gen_tables.py --n 8192 --out gens_8192.bin [--show-style 0|1] [--sqrt-branch 0|1]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bulletproofs_amd import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192, help="generator pairs (power of two, <= 8192)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--show-style", type=int, default=0)
    ap.add_argument("--sqrt-branch", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    eng = Engine(a.device)
    eng._check(eng.lib.bp_ctx_set_show_style(eng.h, a.show_style), "bp_ctx_set_show_style")
    eng._check(eng.lib.bp_ctx_set_sqrt_branch(eng.h, a.sqrt_branch), "bp_ctx_set_sqrt_branch")
    t0 = time.perf_counter()
    eng.derive_generators(a.n)
    t1 = time.perf_counter()
    eng.save_generators(a.out)
    print("derived %d generator pairs in %.2f s, wrote %s (%d bytes)" % (a.n, t1 - t0, a.out, os.path.getsize(a.out)))
    eng.close()


if __name__ == "__main__":
    main()
