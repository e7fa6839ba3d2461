#!/usr/bin/env python3
"""Launch-policy check OFF the 256-atom grid: the thresholds of pf_host.cpp: LaunchPolicy (n16_rows_max,
n16_fuse_rows_max, ...) were swept on synthetic 256-atom pockets; this runs the non-uniform pocket mix of
tests/test_gpu_fullsize.py (pockets of 160 ... 475 atoms, 3 ... 8 centers) at batch 32 and 128 under the DEFAULT policy and
under each FORCED kernel family, and prints sample-steps/s of 100 denoising steps at the end of a T = 500 schedule:

    default                the policy as shipped
    n16 (+ fused launch)   PFDYN_N16=7 PFDYN_N16_ROWS_MAX=<huge>     16-row items on four waves for every conv layer
    n16, no fused launch   PFDYN_N16=3 PFDYN_N16_ROWS_MAX=<huge>
    row-group              PFDYN_N16=0                               4 / 8 rows per wave by the row-group thresholds

The policy "holds" where default >= 0.97 x the best forced family.  Output: profiles/r04/policy_nonuniform.txt (via gpurun).
    python tools/policy_check.py [--steps 100]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pharmacoforge_amd as pfa  # noqa: E402
from pharmacoforge_amd import schedule, synthetic  # noqa: E402

FORMS = [("default", {}),
         ("n16 + fused", {"PFDYN_N16": "7", "PFDYN_N16_ROWS_MAX": "100000000"}),
         ("n16, not fused", {"PFDYN_N16": "3", "PFDYN_N16_ROWS_MAX": "100000000"}),
         ("row-group", {"PFDYN_N16": "0"})]


def make_batch(B, dev, uniform=None):
    sizes_cycle = [3, 3, 4, 4, 5, 5, 6, 6, 7, 8]
    n_prot = [uniform or (160 + (315 * i) // max(B - 1, 1)) for i in range(B)]          # 160 ... 475 atoms
    xs, hs = zip(*[synthetic.synthetic_pocket(7000 + i, n_prot[i]) for i in range(B)])
    px, ph = torch.cat(xs).to(dev), torch.cat(hs).to(dev)
    pptr = torch.tensor([0] + list(__import__("itertools").accumulate(n_prot)), dtype=torch.int64)
    nf = [6 if uniform else sizes_cycle[i % len(sizes_cycle)] for i in range(B)]
    fptr = torch.tensor([0] + list(__import__("itertools").accumulate(nf)), dtype=torch.int64)
    return px, ph, pptr, fptr


def run(B, steps, env, dev, uniform=None):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng = pfa.PfEngine(device=dev)                     # the policy is read from the environment when the handle is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    eng.load_state_dict(synthetic.make_state_dict(0))
    px, ph, pptr, fptr = make_batch(B, dev, uniform)
    s, d = eng.build_pp_edges(px, pptr)
    eng.set_batch(px, ph, pptr, fptr, s, d)
    T = 500
    coef = schedule.step_coefficients(schedule.PredefinedNoiseSchedule('polynomial_2', T, 1e-5).gamma, T)
    n = steps + 20
    carr = eng.coef_array(coef, list(range(n - 1, -1, -1)))
    eng.prepare_timesteps(carr, n)
    Nf = int(fptr[-1])
    noise = torch.randn(n + 1, Nf, 9, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    best = None
    for rep in range(3):
        eng.sample_begin(noise[0])
        for i in range(20):
            eng.denoise_step(carr[i], noise[i + 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20, n):
            eng.denoise_step(carr[i], noise[i + 1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    fam = (eng.kernel_family(0), eng.kernel_family(1), eng.l0_hoist())
    return B * steps / best, best / steps * 1e6, fam, int(pptr[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    run(8, 20, {}, dev)                                    # clocks, first-use set-up
    print("workload                         form              sample-steps/s   us/step   families (conv 0, conv 1, hoist rows)   vs best forced")
    for label, B, uniform in (("256-atom pockets x 6 centers", 32, 256), ("160..475 atoms, 3..8 centers", 32, None),
                              ("160..475 atoms, 3..8 centers", 128, None), ("256-atom pockets x 6 centers", 128, 256)):
        rows = [(name,) + run(B, args.steps, env, dev, uniform) for name, env in FORMS]
        best_forced = max(r[1] for r in rows[1:])
        for name, v, us, fam, natoms in rows:
            tag = f"{v / best_forced:5.2f}" + ("  <- policy" if name == "default" else "")
            print(f"{label:30s} B={B:<4d} {name:16s} {v:12.0f} {us:10.1f}   {str(fam):38s} {tag}")
        print()


if __name__ == "__main__":
    main()
