#!/usr/bin/env python3
"""Headline benchmark: spin-flip attempts / second on the 3D 16^3 +-J glass,
24 temperatures x 4 replicas x 128 disorder realisations per GPU (BASELINE.json configs[2],
SURVEY.md section 8d).

    python bench.py --gpus 1 --steps 2000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full sweep of the protocol over the rank's ensemble: Metropolis sweep of every
system, energies + magnetisations, replica-overlap measurement, running statistics, and one
parallel-tempering event per replica ladder.  State is resident in HBM before the timed region.
Disorder realisations shard over ranks with no data-path collective (weak scaling, 128 per GPU);
the only exchange is the final all-gather of the per-realisation observable blocks.

torch is used for the process group / RCCL all-gather only; the hot path is libpeapods_amd.so.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHAPE = (16, 16, 16)
N_TEMPS, N_REPLICAS, N_DISORDER_PER_GPU = 24, 4, 128
T_MIN, T_MAX = 0.8, 1.4          # reference tests/spin_glass_crossings.py:16
ROOT_SEED = 42                   # reference tests/spin_glass_crossings.py:11
BYTES_PER_ATTEMPT = 0.625        # N/8 read + N/8 write spins + 3N/8 coupling signs
BYTES_PER_ATTEMPT_OVERLAP = 0.125  # 2N/8 read per pair and recorded sweep, 2 pairs per 4 replicas
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--disorder-per-gpu", type=int, default=N_DISORDER_PER_GPU)
    ap.add_argument("--pt-schedule", default="single_random_edge")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--profile-sweeps", type=int, default=64)
    return ap.parse_args()


def build_ensemble(rank, n_per_gpu, world):
    from peapods_amd import IsingSimulation
    from peapods_amd.spin_models import generate_couplings, seed_material

    coupling_seq, dynamics_seed = seed_material(ROOT_SEED)
    mine = range(rank * n_per_gpu, (rank + 1) * n_per_gpu)
    couplings = np.stack(generate_couplings("bimodal", SHAPE, 3, n_per_gpu * world, coupling_seq,
                                            realizations=mine))
    temps = np.linspace(T_MIN, T_MAX, N_TEMPS).astype(np.float32)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sim = IsingSimulation(list(SHAPE), couplings, temps, N_REPLICAS, None, dynamics_seed,
                          device=local_rank, first_realization=mine.start)
    return sim, couplings, temps, dynamics_seed


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes expose every host CPU but grant a share of them)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("PEAPODS_BENCH_THREADS")
    return int(env) if env else cores


def cpu_baseline(couplings, temps, dynamics_seed, schedule, seconds):
    """Times the oracle (the reference's algorithm restated in C: i8 spins, f32 couplings,
    typewriter order, xoshiro256**, integer table) on this box's host cores: one thread per
    disorder realisation, replicas sequential inside, i.e. the reference's outer rayon level."""
    from oracle import bindings as ob

    try:
        lib = ob.load(ob.build(native=True))
    except Exception:
        lib = ob.load()
    cores = host_cores()
    n_real = min(cores, couplings.shape[0])
    sim = ob.OracleSimulation(list(SHAPE), couplings[:n_real], temps, N_REPLICAS, None,
                              dynamics_seed, lib=lib)
    per_sweep = int(np.prod(SHAPE)) * N_TEMPS * N_REPLICAS * n_real
    kw = dict(pt_interval=1, pt_schedule=schedule, warmup_ratio=0.0, n_threads=cores)
    t0 = time.perf_counter()
    sim.run(sim.make_config(2, "metropolis", **kw))
    probe = (time.perf_counter() - t0) / 2
    sweeps = max(4, int(seconds / max(probe, 1e-6)))
    t0 = time.perf_counter()
    sim.run(sim.make_config(sweeps, "metropolis", **kw))
    dt = time.perf_counter() - t0
    return {
        "value": per_sweep * sweeps / dt,
        "unit": "spin-flip attempts/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_real} of {couplings.shape[0]} disorder realisations x {N_TEMPS}T x "
                  f"{N_REPLICAS} replicas of 16^3, {sweeps} sweeps (Metropolis + PT + overlap "
                  f"every sweep), {dt:.1f} s on {cores} threads, gcc -O3 -march=native",
    }


def valu_limiter(kernels_ms, disorder_per_gpu):
    """The resource that actually bounds the two sweep kernels: VALU issue slots.  A wave64 VALU
    instruction holds a SIMD for 4 cycles, so the chip retires 256 CUs x 4 SIMDs x 2.4 GHz / 4
    wave-instructions per second.  Instruction counts per launch are SQ_INSTS_VALU of the committed
    PMC pass (profiles/r01_pmc_per_launch.csv, 128 realisations per launch; they scale with the
    ensemble and do not depend on the data)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_per_launch.csv")
    if not os.path.exists(path):
        return None
    import csv

    counts = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["kernel"]
            if name.startswith("pa::k_rng_overlap"):
                counts["k_rng"] = float(row["SQ_INSTS_VALU"])
            elif name.startswith("pa::k_sweep<"):
                counts["k_sweep"] = float(row["SQ_INSTS_VALU"])
    peak = 256 * 4 * 2.4e9 / 4
    scale = disorder_per_gpu / 128.0
    frac = {k: counts[k] * scale / (kernels_ms[k] * 1e-3) / peak for k in counts if kernels_ms.get(k, 0) > 0}
    return {"resource": "VALU issue (wave64 instructions)", "wave_insts_per_launch": {k: v * scale for k, v in counts.items()},
            "peak_wave_insts_per_s": peak, "frac": frac}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with 1 rank)
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)

    sim, couplings, temps, dynamics_seed = build_ensemble(rank, args.disorder_per_gpu, world)
    n_spins = int(np.prod(SHAPE))
    attempts_per_step = n_spins * N_TEMPS * N_REPLICAS * args.disorder_per_gpu

    def config(n):
        return sim.make_config(n, "metropolis", pt_interval=1, pt_schedule=args.pt_schedule,
                               warmup_ratio=0.0)

    def sync_all():
        sim.synchronize()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

    if args.warmup > 0:
        sim.run(config(args.warmup))
    sync_all()
    t0 = time.perf_counter()
    sim.run(config(args.steps), wait=False)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # final observable exchange: per-realisation blocks, all-gathered over RCCL, aggregated in
    # realisation order (results.rs:165-282) so the result is independent of the GPU count
    t_gather = time.perf_counter()
    from peapods_amd.distributed import aggregate_mean, all_gather_blocks

    blocks = all_gather_blocks(sim.result_blocks(), dist, device="cuda" if dist is not None else None)
    mean = aggregate_mean(blocks)
    gather_ms = (time.perf_counter() - t_gather) * 1e3

    kernels_ms = sim.profile_kernels(config(args.profile_sweeps), args.profile_sweeps)

    if rank == 0:
        total_attempts = attempts_per_step * world * args.steps
        value = total_attempts / elapsed
        sweep_ms = kernels_ms["k_rng"] + kernels_ms["k_sweep"]
        # the random-stream launch also carries the overlap measurement of the previous sweep
        # (k_rng_overlap), so its algorithmic bytes (+0.125 B/attempt) count with it
        alg_bytes = (BYTES_PER_ATTEMPT + BYTES_PER_ATTEMPT_OVERLAP) * attempts_per_step
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        traffic = None
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            with open(traffic_file) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_sweep")
        out = {
            "metric": "spin-flips/sec on 3D 16^3 +-J glass, 24T x 4rep x 128 disorder per GPU",
            "value": value,
            "unit": "spin-flip attempts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-sliced spins (1 bit/spin); u64 xoshiro256**; i32 reductions; f64 sums",
            "data": "synthetic: i.i.d. +-1 couplings (numpy SeedSequence(42) children, as the "
                    "reference generates them), random initial spins",
            "config": {
                "workload": "BASELINE configs[2]: 16^3 +-J, linspace(0.8,1.4,24) x 4 replicas x "
                            f"{args.disorder_per_gpu} disorder per GPU; Metropolis + energies + "
                            "overlap + statistics + parallel tempering every sweep",
                "pt_schedule": args.pt_schedule,
                "update_order": "reference typewriter order and xoshiro256** stream (bit-exact)",
                "attempts_per_step_per_gpu": attempts_per_step,
                "parallelism": f"disorder sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_rng_overlap + k_sweep (the two launches of one Metropolis sweep; the first "
                          "also measures the previous sweep's replica overlap)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": kernels_ms,
                "limiter": valu_limiter(kernels_ms, args.disorder_per_gpu),
            },
            "observable_allgather_ms": gather_ms,
            "checksum_energy_mean": float(mean[3 * N_TEMPS:4 * N_TEMPS].sum()),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(couplings, temps, dynamics_seed, args.pt_schedule,
                                               args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
