"""CPU-only checks of the product's host code through the C ABI: the library loads and exports
every declared symbol, and the host-side building blocks (seed tree, xoshiro jump-ahead,
acceptance table, sweep dependency levels, aggregation, option validation) agree with the
oracle.  No compute call touches a GPU here."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from peapods_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "peapods_amd.h")).read()
    declared = set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", header))
    declared -= {"pa_sim_desc", "pa_sample_config"}
    assert len(declared) >= 35
    raw = C.CDLL(nat.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert declared == set(nat.SIGNATURES), "ctypes binding table out of sync with the header"
    assert nat.lib().pa_version().startswith(b"peapods_amd")
    # ... and nothing else: the diagnostic entry points are declared in their own header, every other pa_* symbol of the
    # dynamic table is in the public one
    import subprocess

    debug = set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "peapods_amd_debug.h")).read()))
    assert debug and all(name.startswith("pa_debug_") for name in debug)
    table = subprocess.run(["nm", "-D", "--defined-only", nat.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in table.splitlines() if line.split()[-1].startswith("pa_")}
    assert exported == declared | debug, sorted(exported ^ (declared | debug))


def test_developer_switches_are_ignored_without_the_gate():
    """Kernel-selection variables count only together with PEAPODS_DEVELOPER=1 (engine.hpp dev_env); what was honoured
    is reported by pa_active_switches.  (A fresh process each: the gate is read once.)"""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); from peapods_amd import _native as n; l = n.lib(); "
            "d = n.PaSimDesc(); import ctypes as C; h = C.c_void_p(); e = C.create_string_buffer(64); "
            "l.pa_sim_create(C.byref(d), C.byref(h), e, 64); print('switches=[' + l.pa_active_switches().decode() + ']')" % ROOT)
    for gate, expected in ((None, "switches=[]"), ("1", "switches=[")):
        env = dict(os.environ, PEAPODS_NO_RELAY="1", PEAPODS_STREAMS="1")
        env.pop("PEAPODS_DEVELOPER", None)
        if gate:
            env["PEAPODS_DEVELOPER"] = gate
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
        assert expected in out, out


def test_abi_layout_and_generated_bindings():
    """The three descriptions of the argument structs -- the header (parsed by tools/gen_bindings.py, which
    writes the Rust binding of INTEGRATION.md), the ctypes binding every GPU test goes through, and the
    compiled library (pa_abi_layout) -- agree field by field, offset by offset (src/lib.rs:176-284 is
    what the structs carry)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_bindings", os.path.join(ROOT, "tools", "gen_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "bindings", "peapods_amd.rs")).read()
    assert committed == gen.generate(), "bindings/peapods_amd.rs is stale: python tools/gen_bindings.py"
    structs = gen.parse_structs(gen.strip_comments(open(gen.HEADER).read()))
    layout = np.zeros(64, dtype=np.uint64)
    n = int(nat.lib().pa_abi_layout(ptr(layout), layout.size))
    assert n <= layout.size
    at = 0
    for cname, binding in (("pa_sim_desc", nat.PaSimDesc), ("pa_sample_config", nat.PaSampleConfig)):
        names = [f[0] for f in binding._fields_]
        assert [f[0] for f in structs[cname]] == names, f"{cname}: header and ctypes fields differ"
        assert int(layout[at]) == C.sizeof(binding), f"sizeof({cname})"
        assert int(layout[at + 1]) == len(names)
        offsets = [int(x) for x in layout[at + 2:at + 2 + len(names)]]
        assert offsets == [getattr(binding, name).offset for name in names], f"{cname}: field offsets"
        for name in names:  # and the generated Rust struct names every field
            assert f"pub {name}:" in committed
        at += 2 + len(names)
    assert at == n
    exported = set(re.findall(r"pub fn (pa_\w+)\(", committed))
    assert exported == set(nat.SIGNATURES)


def test_create_fails_loudly_without_a_device():
    if nat.lib().pa_device_count() > 0:
        pytest.skip("a HIP device is present")
    from peapods_amd import Ising

    with pytest.raises(RuntimeError, match="no HIP device"):
        Ising((4, 4), temperatures=np.array([1.0, 2.0]), seed=1)


def test_seed_tree_matches_oracle(oracle):
    lib, orc = nat.lib(), oracle.load()
    for v in (0, 1, 42, 2**63 + 12345, 2**64 - 1):
        assert lib.pa_splitmix64(v) == orc.orc_splitmix64(v)
        for idx in (0, 1, 95, 1023):
            assert lib.pa_realization_seed(v, idx) == orc.orc_realization_seed(v, idx)
            assert lib.pa_child_seed(v, 0x53A917E14C2D8B6F, idx) == orc.orc_child_seed(
                v, 0x53A917E14C2D8B6F, idx)


def test_xoshiro_state_and_jump_equal_sequential_steps(oracle):
    lib, orc = nat.lib(), oracle.load()
    rng = oracle.make_rngs([987654321])[0]
    start = np.array(list(rng.s), dtype=np.uint64)
    got = np.zeros(4, dtype=np.uint64)
    lib.pa_xoshiro_state(987654321, 0, ptr(got))
    np.testing.assert_array_equal(got, start)
    position = 0
    for n in (1, 2, 63, 64, 247, 256, 4096 - 256, 4096, 100003):
        while position < n:
            orc.orc_rng_next_u64(C.byref(rng))
            position += 1
        want = np.array(list(rng.s), dtype=np.uint64)
        lib.pa_xoshiro_jump(ptr(start), n, ptr(got))
        np.testing.assert_array_equal(got, want, err_msg=f"jump by {n}")
        lib.pa_xoshiro_state(987654321, n, ptr(got))
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("nn", [1, 2, 3])
def test_metropolis_table_matches_oracle(oracle, nn):
    lib, orc = nat.lib(), oracle.load()
    temps = np.array([0.1, 0.7, 0.8, 1.1, 1.4, 2.0, 5.0, 1e6], dtype=np.float32)
    couplings = np.array([-1.0, 0.0, 1.0], dtype=np.float32)
    out = np.zeros(len(temps) * (4 * nn + 1), dtype=np.uint32)
    assert lib.pa_build_metropolis_table(ptr(couplings), 3, ptr(temps), len(temps), nn, ptr(out)) == 0
    handle = orc.orc_lookup_new(ptr(couplings), 3, ptr(temps), len(temps), nn)
    k = 0
    for t in range(len(temps)):
        for e in range(-2 * nn, 2 * nn + 1):
            assert out[k] == orc.orc_lookup_get(handle, t, e)
            k += 1
    orc.orc_lookup_free(handle)
    bad = np.array([0.5, 1.0], dtype=np.float32)
    assert lib.pa_build_metropolis_table(ptr(bad), 2, ptr(temps), len(temps), nn, ptr(out)) == 2
    zero_t = np.array([0.0], dtype=np.float32)
    assert lib.pa_build_metropolis_table(ptr(couplings), 3, ptr(zero_t), 1, nn, ptr(out)) == 2


@pytest.mark.parametrize("nn", [1, 2, 3])
def test_gibbs_table_is_the_heat_bath_rule_on_the_draw_grid(nn):
    """threshold[t][e] must split the 2^24 draws exactly where the reference's per-draw test
    e >= (T/2) ln(u/(1-u)) (sweep.rs:279-282, f32, platform logf) changes from true to false."""
    import ctypes.util

    lib = nat.lib()
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.logf.restype = C.c_float
    libm.logf.argtypes = [C.c_float]
    temps = np.array([0.1, 0.5, 0.8, 1.4, 2.269, 5.0, 50.0], dtype=np.float32)
    couplings = np.array([-1.0, 0.0, 1.0], dtype=np.float32)
    width = 4 * nn + 1
    out = np.zeros(len(temps) * width, dtype=np.uint32)
    assert lib.pa_build_gibbs_table(ptr(couplings), 3, ptr(temps), len(temps), nn, ptr(out)) == 0

    def accepts(t, e, k):
        u = np.float32(k) / np.float32(16777216.0)
        with np.errstate(divide="ignore"):
            ratio = np.float32(u / np.float32(np.float32(1.0) - u))
        logit = np.float32(libm.logf(C.c_float(float(ratio)))) if ratio > 0 else np.float32(-np.inf)
        return np.float32(e) >= np.float32(np.float32(t / np.float32(2.0)) * logit)

    rng = np.random.default_rng(5)
    for ti, t in enumerate(temps):
        for e in range(-2 * nn, 2 * nn + 1):
            thr = int(out[ti * width + e + 2 * nn])
            assert 1 <= thr <= 1 << 24  # u = 0 always flips
            probes = {0, thr - 1, (1 << 24) - 1} | set(int(x) for x in rng.integers(0, 1 << 24, 6))
            if thr < 1 << 24:
                probes.add(thr)
            for k in probes:
                assert bool(accepts(t, e, k)) == (k < thr), (t, e, k, thr)
        row = out[ti * width:(ti + 1) * width]
        assert np.all(np.diff(row.astype(np.int64)) >= 0)  # favourable moves are accepted more often
    bad = np.array([0.5, 1.0], dtype=np.float32)
    assert lib.pa_build_gibbs_table(ptr(bad), 2, ptr(temps), len(temps), nn, ptr(out)) == 2


@pytest.mark.parametrize("shape,offsets", [((16, 16, 16), None), ((5, 7), None), ((2, 3, 4), None),
                                           ((6, 6), [[1, 0], [0, 1], [1, -1]]), ((9,), None)])
def test_sweep_levels_respect_the_typewriter_order(oracle, shape, offsets):
    lib = nat.lib()
    lat = oracle.Lattice(list(shape), offsets=offsets)
    shape_arr = np.asarray(shape, dtype=np.uint64)
    level = np.zeros(lat.n_spins, dtype=np.uint32)
    if offsets is None:
        n_levels = lib.pa_build_sweep_levels(len(shape), ptr(shape_arr), 0, None, ptr(level))
    else:
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n_levels = lib.pa_build_sweep_levels(len(shape), ptr(shape_arr), len(offsets), ptr(off),
                                             ptr(level))
    assert n_levels == level.max() + 1
    for i in range(lat.n_spins):
        for d in range(lat.n_neighbors):
            for j in (lat.neighbor_fwd(i, d), lat.neighbor_bwd(i, d)):
                if j < i:
                    assert level[j] < level[i]  # already updated when i runs
                elif j > i:
                    assert level[j] > level[i]  # still old when i runs
    if offsets is None:  # hyperplanes: level = sum of coordinates
        coords = np.indices(shape).reshape(len(shape), -1)
        np.testing.assert_array_equal(level, coords.sum(axis=0))


def test_aggregation_matches_oracle(oracle):
    lib = nat.lib()
    rng = np.random.default_rng(0)
    per = rng.standard_normal((7, 33))
    mean = np.empty(33)
    lib.pa_aggregate_mean(ptr(per), 7, 33, ptr(mean))
    acc = np.zeros(33)
    for row in per:
        acc = acc + row
    np.testing.assert_array_equal(mean, acc / 7.0)
    one = np.empty(33)
    lib.pa_aggregate_mean(ptr(per), 1, 33, ptr(one))
    np.testing.assert_array_equal(one, per[0])
    hist = rng.integers(0, 1000, size=(5, 12)).astype(np.uint64)
    out = np.empty(12, dtype=np.uint64)
    lib.pa_aggregate_sum_u64(ptr(hist), 5, 12, ptr(out))
    np.testing.assert_array_equal(out, hist.sum(axis=0))


def test_option_validation_messages():
    lib = nat.lib()

    def check(n_replicas=2, **kw):
        cfg = nat.PaSampleConfig()
        cfg.n_sweeps, cfg.warmup_sweeps = 10, 2
        for k, v in kw.items():
            setattr(cfg, k, v)
        err = C.create_string_buffer(256)
        return lib.pa_validate_sample_config(C.byref(cfg), n_replicas, err, 256), err.value.decode()

    assert check()[0] == 0
    assert check(n_sweeps=0) == (1, "n_sweeps must be >= 1")
    assert check(warmup_sweeps=11) == (1, "warmup_sweeps must be <= n_sweeps")
    code, text = check(n_replicas=1, overlap_cluster_interval=1, n_overlap_cluster_modes=1)
    assert code == 1 and text.startswith("overlap cluster requires n_replicas >= max group_size")
    assert check(sweep_mode=1)[0] == 0  # gibbs
    assert check(sweep_mode=2) == (1, "unknown sweep_mode, expected 'metropolis' or 'gibbs'")


def test_python_surface_host_side():
    from peapods_amd._core import _parse_overlap_modes
    from peapods_amd.spin_models import generate_couplings, seed_material

    seq_a, dyn_a = seed_material(41)
    seq_b, dyn_b = seed_material(41)
    assert dyn_a == dyn_b and dyn_a != seed_material(42)[1]
    with pytest.raises(ValueError, match="non-negative integer"):
        seed_material(-1)
    full = generate_couplings("bimodal", (4, 4, 4), 3, 8, seq_a)
    part = generate_couplings("bimodal", (4, 4, 4), 3, 8, seq_b, realizations=range(5, 8))
    for a, b in zip(full[5:], part):
        np.testing.assert_array_equal(a, b)
    assert set(np.unique(full[0])) == {-1.0, 1.0} and full[0].dtype == np.float32
    # sample 0 is stable when the disorder count grows (reference test :45-48)
    np.testing.assert_array_equal(generate_couplings("gaussian", (4, 4), 2, 1, seed_material(7)[0])[0],
                                  generate_couplings("gaussian", (4, 4), 2, 3, seed_material(7)[0])[0])
    assert _parse_overlap_modes("houdayer+jorg") == [1, 2]
    with pytest.raises(ValueError, match="unknown overlap_cluster_build_mode"):
        _parse_overlap_modes("bogus")
    with pytest.raises(ValueError, match="must be even"):
        _parse_overlap_modes("houd3")


def test_shard_ranges_cover_the_ensemble():
    from peapods_amd.distributed import shard_range

    for n_total, world in ((128, 1), (1024, 8), (10, 4), (3, 8)):
        edges = [shard_range(n_total, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n_total
        for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
            assert a1 == b0 and a0 <= a1


def test_fft_autocorrelation_matches_the_ring_sums():
    """host_util.cpp autocorrelation_tau: the FFT backend (zero-padded radix-2 transform) against the exact ring
    sums on a deterministic AR(1)-like series, as the reference checks its own two backends
    (spin-sim/src/statistics/autocorrelation.rs:345-372)."""
    lib = nat.lib()
    rng = np.random.default_rng(3)
    n, cols, lag = 300, 3, 40
    x = np.zeros((n, cols), dtype=np.float32)
    for i in range(1, n):
        x[i] = 0.8 * x[i - 1] + rng.standard_normal(cols).astype(np.float32)
    ring = np.zeros(cols)
    fft = np.zeros(cols)
    lib.pa_autocorrelation_tau(nat.ptr(x), n, cols, lag, nat.ptr(ring), 0)
    lib.pa_autocorrelation_tau(nat.ptr(x), n, cols, lag, nat.ptr(fft), 1)
    np.testing.assert_allclose(fft, ring, rtol=0, atol=1e-10)
    assert np.all(ring > 1.0)  # a correlated series
    const = np.full((8, 1), 3.5, dtype=np.float32)  # degenerate: variance 0 -> gamma = delta, tau = 0.5
    out = np.zeros(1)
    for backend in (0, 1):
        lib.pa_autocorrelation_tau(nat.ptr(const), 8, 1, 4, nat.ptr(out), backend)
        assert out[0] == 0.5


def test_a_rank_that_fails_before_the_gather_fails_every_rank():
    """distributed.sample exchanges an ok / failed word before the first result collective, so a rank whose run raises
    (unsupported configuration, device error, Ctrl-C) does not leave the others blocked in gather_parts."""
    import threading

    from peapods_amd import distributed as dist

    class Stub:  # the three calls distributed.sample makes on a handle
        _lib = None

        def __init__(self, fails):
            self.fails = fails

        def make_config(self, *a, **k):
            return object()

        def run(self, cfg):
            if self.fails:
                raise NotImplementedError("declined on this rank")

        def collect_parts(self, cfg, with_pt):
            return {"blocks": np.zeros((1, 4))}, {}

    group = dist.LoopbackGroup(3)
    errors = [None] * 3

    def rank_main(rank):
        try:
            dist.sample(Stub(rank == 1), group.endpoint(rank), 4, "metropolis")
        except BaseException as exc:
            errors[rank] = exc

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
    assert not any(t.is_alive() for t in threads)
    assert isinstance(errors[1], NotImplementedError)
    assert all(isinstance(errors[r], RuntimeError) and "rank(s) [1]" in str(errors[r]) for r in (0, 2))
