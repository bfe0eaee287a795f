#!/usr/bin/env python3
"""Benchmark of the hot path: spin-flip attempts / second.

    python bench.py                                  # 1 GPU, BASELINE configs[2] (the headline metric)
    python bench.py --gpus 8 --steps 20 --warmup 5   # starts 8 worker processes itself, one per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # also fine
    python bench.py --config c1 | c4                 # BASELINE configs[1] (64^3 ferromagnet) / configs[4]
                                                     # (12^3 glass with the Houdayer move): same JSON shape

One step = one full sweep of the protocol over the rank's ensemble: Metropolis sweep of every system,
energies + magnetisations, replica-overlap measurement, running statistics and one parallel-tempering
event per replica ladder (c4: plus the overlap-cluster move).  State is resident in HBM before the timed
region.  Disorder realisations shard over ranks with no data-path collective (weak scaling, 128 per
GPU); the only exchange is the final all-gather of the per-realisation observable blocks, over RCCL
through the library's own C ABI (pa_comm_*).  No torch anywhere: when launched by torch.distributed.run
only its RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment is read.
"""

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROOT_SEED = 42                   # reference tests/spin_glass_crossings.py:11
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_OVERLAP = 0.125            # 2N/8 read per pair and recorded sweep = 1/8 B per attempt

# shapes, temperature grids and protocols of the BASELINE.json configs.
# bytes = algorithmic bytes per attempt of the sweep (1 bit/spin read + written, 1 sign bit per +-J bond;
# the uniform ferromagnet has no couplings to load).
WORKLOADS = {
    "c2": dict(title="BASELINE configs[2]: 16^3 +-J, linspace(0.8,1.4,24) x 4 replicas x {n} disorder per GPU; "
                     "Metropolis + energies + overlap + statistics + parallel tempering every sweep",
               metric="spin-flips/sec on 3D 16^3 +-J glass, 24T x 4rep x 128 disorder per GPU",
               shape=(16, 16, 16), n_temps=24, n_replicas=4, n_disorder=128, t_range=(0.8, 1.4),
               couplings="bimodal", bytes=0.625, sample={}),
    "c1": dict(title="BASELINE configs[1]: 64^3 ferromagnet, linspace(4.111,4.911,32) x 2 replicas; Metropolis + "
                     "energies + overlap + statistics + parallel tempering every sweep",
               metric="spin-flips/sec on 3D 64^3 ferromagnet, 32T x 2rep",
               shape=(64, 64, 64), n_temps=32, n_replicas=2, n_disorder=1, t_range=(4.111, 4.911),
               couplings="ferro", bytes=0.25, sample={}),
    "c4": dict(title="BASELINE configs[4]: 12^3 +-J, linspace(0.8,1.4,24) x 4 replicas x {n} disorder per GPU; the "
                     "configs[2] protocol + the {mode} overlap-cluster move every sweep",
               metric="spin-flips/sec on 3D 12^3 +-J glass with the overlap-cluster move, 24T x 4rep x 128 disorder per GPU",
               shape=(12, 12, 12), n_temps=24, n_replicas=4, n_disorder=128, t_range=(0.8, 1.4),
               couplings="bimodal", bytes=0.625, sample=dict(overlap_cluster_update_interval=1)),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--disorder-per-gpu", type=int, default=None)
    ap.add_argument("--pt-schedule", default="single_random_edge")
    ap.add_argument("--oc-mode", default="houdayer", help="c4: overlap_cluster_build_mode (houdayer, jorg, cmr)")
    ap.add_argument("--oc-variant", default="wolff", help="c4: overlap_cluster_mode (wolff, sw)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--profile-sweeps", type=int, default=64)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed call of exactly --steps steps is made this many times; the median is reported")
    ap.add_argument("--settle-ms", type=float, default=30.0,
                    help="untimed calls of --steps steps are made back to back for this long before the timed ones")
    ap.add_argument("--settle-calls", type=int, default=64, help="... and at most this many of them")
    ap.add_argument("--no-step-profile", action="store_true",
                    help="skip the per-kernel step profile (a second, one-stream handle): profiler runs want the timed protocol only")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short measurements of BASELINE configs[1] and configs[4] after the headline")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts N workers before anything touches the GPU
# ------------------------------------------------------------------------------------------------
def count_gpus_without_touching_them():
    """GPU agents in the KFD topology (sysfs): nodes with SIMDs.  The launcher must not initialise HIP --
    its children are started by fork + exec."""
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    count = 0
    for path in nodes:
        try:
            with open(path) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
        except (OSError, ValueError):
            pass
    if nodes:
        # a container may expose fewer devices than the host has
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            if os.environ.get(var):
                count = min(count, len([x for x in os.environ[var].split(",") if x.strip() != ""]))
        return count
    # no sysfs view: ask a child process (the parent stays clean)
    probe = subprocess.run([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r); from peapods_amd import _native as n; "
                            "print(n.lib().pa_device_count())" % ROOT], capture_output=True, text=True)
    try:
        return int(probe.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_workers(args):
    have = count_gpus_without_touching_them()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 3
    port = free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:  # one rank failed: the others would wait for it forever
                    other.terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------
# worker
# ------------------------------------------------------------------------------------------------
def build_ensemble(work, rank, n_per_gpu, world, device, stream_groups=0):
    from peapods_amd import IsingSimulation
    from peapods_amd.spin_models import generate_couplings, seed_material

    coupling_seq, dynamics_seed = seed_material(ROOT_SEED)
    mine = range(rank * n_per_gpu, (rank + 1) * n_per_gpu)
    couplings = np.stack(generate_couplings(work["couplings"], work["shape"], 3, n_per_gpu * world, coupling_seq,
                                            realizations=mine))
    temps = np.linspace(*work["t_range"], work["n_temps"]).astype(np.float32)
    sim = IsingSimulation(list(work["shape"]), couplings, temps, work["n_replicas"], None, dynamics_seed,
                          device=device, first_realization=mine.start, stream_groups=stream_groups)
    return sim, couplings, temps, dynamics_seed


def step_profile(work, sample_kw, n_per_gpu, device, n_sweeps=48, settle=200):
    """Every launch of one step of the timed protocol, by kernel (pa_sim_profile_step): a second handle of the same
    ensemble with ONE stream group -- a launch's time beside another group's launches says little about the launch --
    runs sample()'s own launch sequence with an event in front of every launch.  {kernel: {ms_per_step,
    launches_per_step}}, and the step time of that one-stream handle."""
    sim, *_ = build_ensemble(work, 0, n_per_gpu, 1, device, stream_groups=1)
    cfg = sim.make_config(n_sweeps, "metropolis", warmup_ratio=0.0, **sample_kw)
    sim.run(sim.make_config(settle, "metropolis", warmup_ratio=0.0, **sample_kw))
    table, step_ms = sim.profile_step(cfg, n_sweeps)
    rows = {name: {"ms_per_step": ms, "launches_per_step": launches} for name, (ms, launches) in table.items()}
    return rows, step_ms


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


def cpu_baseline(work, couplings, temps, dynamics_seed, sample_kw, seconds):
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
    sim = ob.OracleSimulation(list(work["shape"]), couplings[:n_real], temps, work["n_replicas"], None,
                              dynamics_seed, lib=lib)
    per_sweep = int(np.prod(work["shape"])) * work["n_temps"] * work["n_replicas"] * n_real
    kw = dict(sample_kw, warmup_ratio=0.0, n_threads=cores)
    t0 = time.perf_counter()
    sim.run(sim.make_config(2, "metropolis", **kw))
    probe = (time.perf_counter() - t0) / 2
    sweeps = max(4, int(seconds / max(probe, 1e-6)))
    t0 = time.perf_counter()
    sim.run(sim.make_config(sweeps, "metropolis", **kw))
    dt = time.perf_counter() - t0
    side = "x".join(str(s) for s in work["shape"])
    return {
        "value": per_sweep * sweeps / dt,
        "unit": "spin-flip attempts/s",
        "cores": min(cores, n_real),
        "kind": "port",
        "sample": f"{n_real} of {couplings.shape[0]} disorder realisations x {work['n_temps']}T x "
                  f"{work['n_replicas']} replicas of {side}, {sweeps} sweeps of the same protocol, "
                  f"{dt:.1f} s on {min(cores, n_real)} threads, gcc -O3 -march=native",
    }


def load_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


def library_source_hash():
    """Hash of the sources libpeapods_amd.so was built from, as the library itself reports it (pa_version)."""
    from peapods_amd import _native as nat

    text = nat.lib().pa_version().decode()
    return text.split("src:")[-1] if "src:" in text else None


def current_counters():
    """profiles/counters.json, or (None, reason) when it was collected from other kernel sources than the ones
    loaded now: the PMC figures describe a binary, not a file name."""
    counters = load_json("counters.json")
    if not counters:
        return None, "profiles/counters.json is missing"
    have, want = counters.get("source_hash"), library_source_hash()
    if have is None or want is None or have != want:
        return None, (f"profiles/counters.json was collected from kernel sources {have} (commit "
                      f"{counters.get('commit')}), the loaded library was built from {want}: recollect with "
                      "tools/run_collect.sh")
    return counters, None


def valu_limiter(kernels_ms, scale, step_ms=None):
    """What actually bounds the two sweep kernels: VALU issue, priced against BOTH peaks.
    (a) the hardware guide's figure (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles, 4 SIMDs
        per CU, 256 CUs) at the clock the probe observed under this load;
    (b) the rate the probe measured for the generator step itself (tools/probe/valu_issue.hip ->
        profiles/r02_valu_issue.json: xoshiro256** + 3 compares + ballot parking at 8 waves per SIMD; the step's
        instruction mix does not reach (a): tools/probe/step_variants.hip, profiles/r03_step_variants.json).
    Wave-instruction counts per launch are SQ_INSTS_VALU of the committed PMC pass (they scale with the ensemble
    and do not depend on the data); null when that pass was collected from other kernel sources."""
    counters, stale = current_counters()
    probe = load_json("r03_valu_issue.json") or load_json("r02_valu_issue.json")
    if not probe:
        return None
    if not counters:
        return {"resource": "VALU issue (wave64 instructions)", "wave_insts_per_launch": None, "reason": stale}
    rate, clock = {}, {}
    for row in probe["rows"]:
        if row["waves_per_simd"] == 8:
            rate[row["op"]] = row["wall_inst_per_s"]
            clock[row["op"]] = row["clock_mhz"]
    counts = {k: v * scale for k, v in counters.get("wave_insts_per_launch", {}).items()}
    peak_probe = rate.get("rng_step_full")
    peak_guide = 256 * 4 * clock.get("rng_step_full", 2400) * 1e6 / 2.0  # 2 cycles per wave-instruction per SIMD
    def fracs(peak):
        return {k: counts[k] / (kernels_ms[k] * 1e-3) / peak for k in counts if kernels_ms.get(k, 0) > 0 and peak}
    return {"resource": "VALU issue (wave64 instructions)", "wave_insts_per_launch": counts,
            "counters_collected_at": counters.get("commit"), "counters_source_hash": counters.get("source_hash"),
            "peak_guide_wave_insts_per_s": peak_guide,
            "peak_guide_source": "MI355X_MICROARCH.md: 2 cycles per wave64 VALU instruction per SIMD, 1024 SIMDs, at the "
                                 f"{clock.get('rng_step_full', 2400)} MHz the probe observed",
            "frac_of_guide_peak": fracs(peak_guide),
            "peak_probe_wave_insts_per_s": peak_probe,
            "peak_probe_source": "profiles/r03_valu_issue.json: rng_step_full at 8 waves/SIMD (tools/probe/valu_issue.hip; disassembly: tools/probe/valu_issue_excerpt.s)",
            "frac_of_probe_peak": fracs(peak_probe),
            # both kernels of a sweep over the STEP of the timed run (two stream groups side by side): how busy the vector
            # units are over a whole step, not inside one launch
            "frac_of_guide_peak_step": (sum(counts.values()) / (step_ms * 1e-3) / peak_guide) if step_ms and counts else None,
            "frac_of_probe_peak_step": (sum(counts.values()) / (step_ms * 1e-3) / peak_probe) if step_ms and counts and peak_probe else None,
            "steady_kernels": counters.get("steady_kernels")}


def roofline_of(work, attempts, ms_per_step, pair_ms, kernels_per_step, one_stream_step_ms, note):
    """The roofline block of a config: `step` prices the WHOLE step (ms_per_step of the timed run: every launch, the
    cluster move of c4 included), `pair` the random stream + sweep launches alone (what rounds 1-3 reported as frac)."""
    alg = (work["bytes"] + BYTES_OVERLAP) * attempts
    step_gbs = alg / (ms_per_step * 1e-3) / 1e9 if ms_per_step > 0 else 0.0
    pair_gbs = alg / (pair_ms * 1e-3) / 1e9 if pair_ms and pair_ms > 0 else None
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_step": alg,
            "achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
            "frac_is": "algorithmic bytes of one step / ms_per_step of the timed run (every launch of the step)",
            "pair": {"achieved": pair_gbs, "frac": pair_gbs / HBM_PEAK_GBS if pair_gbs else None, "ms": pair_ms,
                     "kernels": "the random-stream launch + the sweep launch of one Metropolis sweep"},
            "kernels_per_step": kernels_per_step, "one_stream_step_ms": one_stream_step_ms, "note": note}


def pair_ms_of(kernels_per_step):
    """Random stream + sweep of one Metropolis sweep out of a step profile (names as written at the launch sites)."""
    if not kernels_per_step:
        return None
    total = 0.0
    for name, row in kernels_per_step.items():
        base = name.split("<")[0]
        if base in ("k_rng", "k_rng_relay", "k_rng_overlap", "k_rng_f") or base.startswith("k_sweep"):
            total += row["ms_per_step"]
    return total or None


def other_configs(args, device):
    """configs[1] (64^3 ferromagnet) and configs[4] (12^3 glass with the Houdayer / Joerg overlap move, single-cluster and
    union-find forms) on this GPU: value and ms_per_step from a warm-up call and one timed call each, then every launch of
    a step by kernel from a one-stream handle (step_profile), and the roofline fraction of the step and of the sweep pair."""
    out = {}
    c4 = lambda mode, variant: dict(overlap_cluster_build_mode=mode, overlap_cluster_mode=variant)
    todo = [("c1", "c1", {}, 300), ("c4_houdayer", "c4", c4("houdayer", "wolff"), 300), ("c4_jorg", "c4", c4("jorg", "wolff"), 100),
            ("c4_houdayer_sw", "c4", c4("houdayer", "sw"), 200), ("c4_jorg_sw", "c4", c4("jorg", "sw"), 200)]
    for name, key, extra, steps in todo:
        work = WORKLOADS[key]
        try:
            sim, _couplings, _temps, _seed = build_ensemble(work, 0, work["n_disorder"], 1, device)
            kw = dict(work["sample"], pt_interval=1, pt_schedule=args.pt_schedule, **extra)
            attempts = int(np.prod(work["shape"])) * work["n_temps"] * work["n_replicas"] * work["n_disorder"]
            sim.run(sim.make_config(max(steps // 5, 10), "metropolis", warmup_ratio=0.0, **kw))
            sim.synchronize()
            t0 = time.perf_counter()
            sim.run(sim.make_config(steps, "metropolis", warmup_ratio=0.0, **kw), wait=False)
            sim.synchronize()
            dt = time.perf_counter() - t0
            del sim
            try:
                kernels, one_stream = step_profile(work, kw, work["n_disorder"], device, n_sweeps=32, settle=max(steps // 5, 10))
            except Exception as exc:
                kernels, one_stream = None, f"{type(exc).__name__}: {exc}"
            out[name] = {"workload": work["title"].format(n=work["n_disorder"], mode="/".join(extra.values()) or "-"),
                         "value": attempts * steps / dt, "unit": "spin-flip attempts/s", "steps": steps,
                         "ms_per_step": dt / steps * 1e3,
                         "roofline": roofline_of(work, attempts, dt / steps * 1e3, pair_ms_of(kernels), kernels, one_stream,
                                                 "kernels_per_step: one stream group; ms_per_step: the default two")}
        except Exception as exc:  # the headline line must not be lost to a side measurement
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def worker(args, rank, world, local_rank):
    from peapods_amd import _native as nat
    from peapods_amd.distributed import Comm, aggregate_mean

    work = WORKLOADS[args.config]
    n_per_gpu = args.disorder_per_gpu or work["n_disorder"]
    # (PEAPODS_BENCH_FORCE_COMM: take the RCCL path with one rank too -- how a 1-GPU box rehearses the 8-GPU run)
    comm = Comm(rank, world, local_rank) if world > 1 or os.environ.get("PEAPODS_BENCH_FORCE_COMM") else None

    sim, couplings, temps, dynamics_seed = build_ensemble(work, rank, n_per_gpu, world, local_rank)
    n_spins = int(np.prod(work["shape"]))
    attempts_per_step = n_spins * work["n_temps"] * work["n_replicas"] * n_per_gpu
    sample_kw = dict(work["sample"], pt_interval=1, pt_schedule=args.pt_schedule)
    if "overlap_cluster_update_interval" in sample_kw:
        sample_kw.update(overlap_cluster_build_mode=args.oc_mode, overlap_cluster_mode=args.oc_variant)

    def config(n):
        return sim.make_config(n, "metropolis", warmup_ratio=0.0, **sample_kw)

    def sync_all():
        sim.synchronize()
        if comm is not None:
            comm.barrier()

    # attainable HBM bandwidth of this device, measured in this run -- before the timed region, so
    # that the short default run of the driver (20 steps = 1.5 ms) does not start on an idle chip's clocks
    copy_gbs = None
    if True:  # every rank: the ranks stay in step
        import ctypes as C

        out_bw, err = C.c_double(), C.create_string_buffer(256)
        if nat.lib().pa_device_copy_bandwidth(local_rank, 1 << 30, 8, C.byref(out_bw), err, len(err)) == nat.PA_OK:
            copy_gbs = out_bw.value
    if args.warmup > 0:
        sim.run(config(args.warmup))
    # The timed call -- EXACTLY args.steps steps between barrier + synchronise on both sides, max over ranks -- is made
    # `repeats` times and the median is reported: a 20-step call lasts 1.3 ms, and one such sample scatters by several
    # per cent with the host's scheduling (profiles/r02_fixed_cost.json).  Every repeat is a complete, separately
    # bracketed measurement of the same K steps; nothing is skipped or carried between them.
    repeats = max(1, args.repeats)
    timings = []
    timed_config = config(args.steps)  # (argument checking is not part of a step: built once, outside the timed region)
    # Settling: a call of 20 steps lasts 1.2 ms, and on a chip that has just been idle it runs 12 % slower than the same
    # call in a train of such calls -- 70 -> 62 us per step over the first fifteen back-to-back calls (20 ms), and back to
    # 68 after every 2 ms of idling (profiles/r03_settle_series.json: the clocks follow the load).  Untimed calls of the
    # same K steps are therefore made back to back until --settle-ms of them have run; they are reported, and nothing of
    # the timed calls depends on them (each timed call is bracketed and synchronised by itself).
    # (... and until the last of them is within 2 % of the fastest: a call that is held up for milliseconds -- seen once in
    # twenty collections, 6 ms in the middle of the train -- leaves the next ones 5 % slow for a while.)
    settle, settled = [], 0.0
    while len(settle) < max(0, args.settle_calls) and (settled < args.settle_ms * 1e-3 or settle[-1] > 1.02 * min(settle)):
        sync_all()
        t0 = time.perf_counter()
        sim.run(timed_config, wait=False)
        sim.synchronize()
        settle.append(time.perf_counter() - t0)
        if comm is not None:
            settle[-1] = comm.max_f64(settle[-1])  # every rank makes the same number of calls
        settled += settle[-1]
    for _ in range(repeats):
        sync_all()
        t0 = time.perf_counter()
        sim.run(timed_config, wait=False)
        enqueue = (time.perf_counter() - t0) * 1e3
        sim.synchronize()
        own = time.perf_counter() - t0
        if comm is not None:
            comm.barrier()
        elapsed = time.perf_counter() - t0
        if comm is not None:
            elapsed = comm.max_f64(elapsed)
            own_all = comm.all_gather(np.array([own])).reshape(-1)
        else:
            own_all = np.array([own])
        timings.append((elapsed, enqueue, own_all))
    order = sorted(range(repeats), key=lambda k: timings[k][0])
    elapsed, enqueue_ms, own_all = timings[order[(repeats - 1) // 2]]  # the median call (the lower one of an even count)
    per_rank = (attempts_per_step * args.steps / own_all).tolist()

    # final observable exchange: per-realisation blocks, all-gathered over RCCL, aggregated in
    # realisation order (results.rs:165-282) so the result is independent of the GPU count
    t_gather = time.perf_counter()
    blocks = comm.gather_result_blocks(sim, n_per_gpu * world) if comm is not None else sim.result_blocks()
    gather_ms = (time.perf_counter() - t_gather) * 1e3
    mean = aggregate_mean(blocks)
    # ... and the full result of the sharded run as distributed.sample assembles it on rank 0: every per-realisation
    # array (P(q) histograms and conditional sums: 2.4 MB per realisation at this shape, tempering counters) gathered
    # to the root in realisation order over the same communicator (pa_comm_gatherv)
    full_gather_ms, full_gather_bytes = None, None
    if comm is not None:
        from peapods_amd.distributed import gather_parts

        parts, _meta = sim.collect_parts(config(args.steps), True)
        comm.barrier()
        t_full = time.perf_counter()
        gathered = gather_parts(parts, comm, root=0)
        comm.barrier()
        full_gather_ms = (time.perf_counter() - t_full) * 1e3
        if rank == 0:
            full_gather_bytes = int(sum(a.nbytes for a in gathered.values()))
        del parts, gathered

    kernels_per_step, one_stream_step_ms = None, None
    if rank == 0 and not args.no_step_profile:
        try:  # every launch of the step by kernel -- for c4 this is where the cluster move's launches show
            kernels_per_step, one_stream_step_ms = step_profile(work, sample_kw, n_per_gpu, local_rank,
                                                                n_sweeps=max(8, args.profile_sweeps))
        except Exception as exc:
            one_stream_step_ms = f"{type(exc).__name__}: {exc}"
    if rank == 0:
        value = attempts_per_step * world * args.steps / elapsed
        # the launches of one Metropolis sweep, out of the step profile (one stream group: every launch covers the whole
        # ensemble): random stream (with whatever rides in it), sweep (+ energies), and measurement launches of their own
        kernels_ms = {"k_rng": 0.0, "k_sweep": 0.0, "k_measure": 0.0}
        for name, row in (kernels_per_step or {}).items():
            base = name.split("<")[0]
            if base.startswith("k_rng"):
                kernels_ms["k_rng"] += row["ms_per_step"]
            elif base.startswith("k_sweep") or base.startswith("k_energy"):
                kernels_ms["k_sweep"] += row["ms_per_step"]
            elif base.startswith("k_measure") or base.startswith("k_overlap"):
                kernels_ms["k_measure"] += row["ms_per_step"]
        sweep_ms = kernels_ms["k_rng"] + kernels_ms["k_sweep"]
        # the random-stream launch also carries the overlap measurement of the previous sweep
        # (k_rng_overlap), so its algorithmic bytes (+0.125 B/attempt) count with it
        alg_bytes = (work["bytes"] + BYTES_OVERLAP) * attempts_per_step
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        counters, counters_stale = current_counters() if args.config == "c2" else (None, "collected for config c2 only")
        scale = n_per_gpu / 128.0
        out = {
            "metric": work["metric"],
            "value": value,
            "unit": "spin-flip attempts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": repeats,
            "timing": f"median of {repeats} separately bracketed calls of exactly {args.steps} steps",
            "ms_per_step_all_repeats": [t[0] / args.steps * 1e3 for t in timings],
            "untimed_settling_calls": len(settle),
            "untimed_settling_ms_per_step": [t / args.steps * 1e3 for t in settle],
            "host_enqueue_ms": enqueue_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-sliced spins (1 bit/spin); u64 xoshiro256**; i32 reductions; f64 sums",
            "data": f"synthetic: {'i.i.d. +-1' if work['couplings'] == 'bimodal' else 'uniform +1'} couplings "
                    "(numpy SeedSequence(42) children, as the reference generates them), random initial spins",
            "config": {
                "workload": work["title"].format(n=n_per_gpu, mode=f"{args.oc_mode}/{args.oc_variant}"),
                "pt_schedule": args.pt_schedule,
                "update_order": "reference typewriter order and xoshiro256** stream (bit-exact)",
                "attempts_per_step_per_gpu": attempts_per_step,
                "parallelism": f"disorder sharded over {world} GPU(s), no data-path collective",
            },
            "ranks": {"launched": world, "rccl_ranks": comm.world if comm is not None else 1,
                      "attempts_per_s_per_rank": per_rank},
            "roofline": {
                "bound": "hbm",
                "kernel": "k_rng_relay + k_sweep_hc (the two launches of one Metropolis sweep; the first also "
                          "carries the previous sweep's overlap measurement, statistics and tempering event)"
                          if args.config != "c1" else
                          "k_rng + k_sweep_slab (random stream + LDS-resident slab sweep of one Metropolis sweep)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "attainable_copy_gbs": copy_gbs,
                "frac_of_attainable": achieved / copy_gbs if copy_gbs else None,
                "traffic": counters.get("hbm_bytes_per_sweep") * scale if counters else None,
                "traffic_collected_at": counters.get("commit") if counters else None,
                "traffic_null_reason": None if counters else counters_stale,
                "algorithmic_bytes_per_launch": alg_bytes,
                "step": {"achieved": alg_bytes / (elapsed / args.steps) / 1e9, "frac": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "is": "the same algorithmic bytes over ms_per_step of the timed run (every launch of the step, "
                               "two stream groups side by side)"},
                "kernels_per_step": kernels_per_step,
                "one_stream_step_ms": one_stream_step_ms,
                "kernel_ms": kernels_ms,
                "kernel_ms_protocol": "sums over kernels_per_step (one stream group): k_rng* | k_sweep*, k_energy* | k_measure*, k_overlap*",
                "limiter": valu_limiter(kernels_ms, scale, elapsed / args.steps * 1e3) if args.config == "c2" else None,
            },
            "observable_allgather_ms": gather_ms,
            "full_result_gather_ms": full_gather_ms,
            "full_result_gather_bytes": full_gather_bytes,
            "library": nat.lib().pa_version().decode(),
            "developer_switches": nat.lib().pa_active_switches().decode(),  # "" unless PEAPODS_DEVELOPER=1 + switches
            "checksum_energy_mean": float(mean[3 * work["n_temps"]:4 * work["n_temps"]].sum()),
        }
        if args.config == "c2" and world == 1 and not args.no_other_configs:
            # the other BASELINE configs, measured in this same run after the headline's timed region (short:
            # well under a second of GPU time each; their full runs are `--config c1`, `--config c4 --oc-mode ...`)
            del sim
            out["other_configs"] = other_configs(args, local_rank)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(work, couplings, temps, dynamics_seed,
                                               {k: v for k, v in sample_kw.items()}, args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


def main():
    args = parse_args()
    if "RANK" in os.environ:  # a worker: started by launch_workers below or by torch.distributed.run
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
            sys.exit(2)
        worker(args, int(os.environ["RANK"]), world, int(os.environ.get("LOCAL_RANK", os.environ["RANK"])))
    elif args.gpus > 1:
        sys.exit(launch_workers(args))
    else:
        worker(args, 0, 1, 0)


if __name__ == "__main__":
    main()
