"""GPU parity of the statistics kernel and the device rings against the CPU oracle and the golden
vectors from the real reference.  All calls go through the C ABI (ctypes)."""
import ctypes

import numpy as np
import pytest
import torch

import synth
from oracle import oracle
from util import close, load_golden

pytestmark = pytest.mark.gpu

STATS = ("MIN", "MAX", "MED", "AVG", "STD", "NUM")


@pytest.fixture(scope="module")
def be():
    from nvrx_straggler.backend import get_backend

    return get_backend()


def _run(be, samples, counts, kinds=None):
    s = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32)).cuda()
    c = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int32)).cuda()
    k = torch.from_numpy(np.ascontiguousarray(kinds, dtype=np.uint8)).cuda() if kinds is not None else None
    return be.row_stats(s, c, k).cpu().numpy()


def _check_against_oracle(got, samples, counts, kinds, tag=""):
    exp = oracle.rows_stats(samples, counts, kinds)
    for r in range(samples.shape[0]):
        n = int(counts[r])
        kind = int(kinds[r]) if kinds is not None else 0
        if n == 0:
            assert np.isnan(got[r, :5]).all() and got[r, 5] == 0, (tag, r)
            continue
        # selections: bit-exact on the f32 samples
        assert got[r, 0] == np.float32(exp[r, 0]), (tag, r, "MIN")
        assert got[r, 1] == np.float32(exp[r, 1]), (tag, r, "MAX")
        assert got[r, 2] == np.float32(exp[r, 2]), (tag, r, "MED", n, kind, got[r, 2], exp[r, 2])
        assert got[r, 5] == n
        # accumulations: f64 on the device, rounded to f32 on output.  The reference's kernel-row
        # path accumulates sequentially in f32 (CuptiProfiler.cpp:63-69) and is itself only accurate
        # to ~n*eps, hence the looser bound for kind 1.
        tol = 1e-6 if kind == 0 else 2e-4
        assert close(got[r, 3], exp[r, 3], rel=tol, abs_=1e-30), (tag, r, "AVG", got[r, 3], exp[r, 3])
        if n > 1 or kind == 1:
            assert close(got[r, 4], exp[r, 4], rel=max(tol, 2e-6), abs_=1e-6 * abs(exp[r, 3]) + 1e-30), (tag, r, "STD", got[r, 4], exp[r, 4])
        else:
            assert np.isnan(got[r, 4])


def _golden_matrix():
    cases = synth.section_stat_cases()
    kept = [synth.retained(c["values"]) for c in cases]
    stride = 8192
    m = np.zeros((len(kept), stride), dtype=np.float32)
    counts = np.zeros(len(kept), dtype=np.uint32)
    for i, v in enumerate(kept):
        m[i, : v.size] = v
        counts[i] = v.size
    return cases, m, counts


def test_section_rows_match_reference_golden(be):
    """kind 0 vs outputs of the reference's Detector._get_section_summaries (section_stats.json)."""
    cases, m, counts = _golden_matrix()
    got = _run(be, m, counts)
    g = {c["name"]: c for c in load_golden("section_stats.json")["cases"]}
    for i, c in enumerate(cases):
        e = g[c["name"]]["expected"]
        assert got[i, 0] == np.float32(e["MIN"]) and got[i, 1] == np.float32(e["MAX"])
        assert got[i, 2] == np.float32(e["MED"]), (c["name"], got[i, 2], e["MED"])
        assert got[i, 5] == e["NUM"]
        assert close(got[i, 3], e["AVG"], rel=1e-6), c["name"]
        assert close(got[i, 4], e["STD"], rel=2e-6, abs_=1e-6 * abs(e["AVG"])), (c["name"], got[i, 4], e["STD"])
    _check_against_oracle(got, m, counts, None, "golden-k0")


def test_kernel_rows_match_reference_golden(be):
    """kind 1 vs outputs of the reference's computeStats (native.json)."""
    cases, m, counts = _golden_matrix()
    kinds = np.ones(len(cases), dtype=np.uint8)
    # computeStats golden was taken on the full pushed vectors; rings hold the newest 8192
    got = _run(be, m, counts, kinds)
    g = {c["name"]: c for c in load_golden("native.json")["compute_stats"]}
    for i, c in enumerate(cases):
        if c["values"].size > 8192:
            continue
        e = g[c["name"]]["expected"]
        assert got[i, 0] == np.float32(e[0]) and got[i, 1] == np.float32(e[1])
        assert got[i, 2] == np.float32(e[2]), (c["name"], "median", got[i, 2], e[2])
        assert got[i, 5] == e[5]
        assert close(got[i, 3], e[3], rel=2e-4), (c["name"], "avg", got[i, 3], e[3])
        assert close(got[i, 4], e[4], rel=2e-3, abs_=2e-4 * abs(e[3])), (c["name"], "std", got[i, 4], e[4])
    _check_against_oracle(got, m, counts, kinds, "golden-k1")


@pytest.mark.parametrize("stride", [4, 8, 64, 256, 1000, 1024, 2048, 3072, 4096, 5000, 8192, 10000, 16384, 20480, 32768, 65536])
def test_random_rows_every_launch_shape(be, stride):
    """Ragged counts (0, 1, 2, odd, even, full) x both kinds for every register-tile variant."""
    rng = np.random.default_rng(stride)
    rows = 24
    m = rng.lognormal(1.0, 0.8, (rows, stride)).astype(np.float32)
    m[3] = np.round(m[3], 1)  # heavy duplicates
    m[4] = 42.0  # all equal
    m[5, ::7] *= 1e4  # outliers stretch the key range
    m[6] = -m[6]  # negative values order correctly
    # rows whose keys are the raw bit patterns until a set sign bit shows up somewhere (cold re-keying path): mixed signs
    # with -0.0, and ONE negative sample at the very end of an otherwise positive row (tile 0 sees none)
    m[9] = rng.normal(0.5, 1.0, stride).astype(np.float32)  # (mean well away from 0: AVG is compared relatively)
    m[9, stride // 2] = -0.0
    m[10, -1] = -m[10, -1]
    # ~100 / ~400 distinct values: the median's bin holds many EQUAL members (ranking by all waves; lanes holding three
    # or more members of the bin: the key-by-key listing)
    m[11] = 10.0 + 0.01 * rng.integers(0, 100, stride)
    m[12] = 10.0 + 0.001 * rng.integers(0, 400, stride)
    counts = rng.integers(0, stride + 1, rows).astype(np.uint32)
    counts[:8] = [stride, max(stride - 1, 0), 1, stride, stride, stride, stride, 2]
    counts[8] = 0
    counts[9:13] = [stride, stride, stride, max(stride - 3, 1)]
    kinds = (np.arange(rows) % 2).astype(np.uint8)
    got = _run(be, m, counts, kinds)
    _check_against_oracle(got, m, counts, kinds, f"stride{stride}")


@pytest.mark.parametrize("stride", [1000, 4096, 4100, 10000, 20000, 33000, 40000, 50000, 65536])
def test_rows_that_defeat_the_range_estimate(be, stride):
    """The kernel bins every key over a range estimated from the row's FIRST tile (THREADS x 4 samples) and clamps what
    falls outside into the two edge bins.  Rows built to make that estimate as wrong as possible -- the first tile a
    constant, a narrow cluster or the low / high end of a sorted row; the rest far away on one or both sides; two
    clusters with the median in the gap; 60 orders of magnitude; denormals and signed zeros; one value different from
    all others -- for both kinds (lower median / mean of the two middles) and odd, even and ragged counts.  Selections
    bit-exact against the oracle."""
    rng = np.random.default_rng(stride)
    n = stride
    first = min(4096, n // 2)
    rows = []

    def add(x):
        rows.append(np.asarray(x, dtype=np.float32))

    x = rng.uniform(100.0, 200.0, n); x[:first] = 1.0; add(x)                      # first tile constant, rest far above
    x = rng.uniform(100.0, 200.0, n); x[:first] = 1e6; add(x)                      # ... far below the first tile
    x = rng.uniform(1.0, 1.001, n); x[first:] = rng.uniform(-1e5, 1e5, n - first); add(x)   # narrow first tile, both sides
    x = np.full(n, 7.25); x[:first] = rng.uniform(0.0, 1e4, first); add(x)         # wide first tile, the rest one value
    add(np.sort(rng.lognormal(0.0, 3.0, n)))                                        # ascending: first tile = the low end
    add(np.sort(rng.lognormal(0.0, 3.0, n))[::-1].copy())                           # descending
    x = np.concatenate([rng.normal(1.0, 0.01, n // 2), rng.normal(1e4, 1.0, n - n // 2)]); add(x)     # median in the gap
    x = np.concatenate([rng.normal(1e4, 1.0, n // 2), rng.normal(1.0, 0.01, n - n // 2)]); add(x)     # ... clusters swapped
    add(10.0 ** rng.uniform(-30.0, 30.0, n))                                        # 60 orders of magnitude
    x = rng.choice(np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, 1e-38], dtype=np.float32), n); add(x)   # denormals, signed zeros
    x = np.full(n, 3.0); x[n // 3] = 2.0; add(x)                                    # one value below all others
    x = np.full(n, 3.0); x[-1] = 4.0; add(x)                                        # ... above, in the last slot
    x = rng.normal(10.0, 0.3, n); x[first:] += np.linspace(0.0, 50.0, n - first); add(x)             # a drifting row
    x = rng.integers(0, 3, n).astype(np.float32); add(x)                            # three distinct values
    m = np.stack(rows)
    R = m.shape[0]
    for kinds, counts in (
        (np.zeros(R, np.uint8), np.full(R, n, np.uint32)),
        (np.ones(R, np.uint8), np.full(R, n, np.uint32)),
        (np.ones(R, np.uint8), np.full(R, n - 1, np.uint32)),
        ((np.arange(R) % 2).astype(np.uint8), rng.integers(first + 1, n + 1, R).astype(np.uint32)),
    ):
        got = _run(be, m, counts, kinds)
        exp = oracle.rows_stats(m, counts, kinds)
        for r in range(R):
            for c, name in ((0, "MIN"), (1, "MAX"), (2, "MED")):
                assert got[r, c] == np.float32(exp[r, c]) or (got[r, c] == 0.0 and exp[r, c] == 0.0), \
                    (stride, r, name, int(kinds[r]), int(counts[r]), got[r, c], exp[r, c])
            assert got[r, 5] == counts[r]


def test_stress_shape_512_rows_x_10000(be):
    """The folded N=1 workload (8 ranks x 64 sections x 10 000 samples): exact medians for all rows."""
    m = np.concatenate([synth.stress_samples(r, 64, 10_000) for r in range(8)], axis=0)
    counts = np.full(512, 10_000, dtype=np.uint32)
    got = _run(be, m, counts)
    _check_against_oracle(got, m, counts, None, "stress")
    # size-independent property: statistics are permutation invariant
    rng = np.random.default_rng(1)
    perm = rng.permutation(10_000)
    got2 = _run(be, m[:, perm], counts)
    assert np.array_equal(got[:, :3], got2[:, :3]) and np.array_equal(got[:, 5], got2[:, 5])
    assert np.allclose(got[:, 3:5], got2[:, 3:5], rtol=1e-6)


def test_linearity_and_shift_properties(be):
    """median(a*x) == a*median(x) for power-of-two a (exact in f32); min<=med<=max; std>=0."""
    rng = np.random.default_rng(9)
    m = rng.normal(10, 0.3, (16, 8192)).astype(np.float32)
    counts = np.full(16, 8192, dtype=np.uint32)
    a = _run(be, m, counts)
    b = _run(be, m * np.float32(4.0), counts)
    assert np.array_equal(a[:, :3] * 4.0, b[:, :3])
    assert (a[:, 0] <= a[:, 2]).all() and (a[:, 2] <= a[:, 1]).all() and (a[:, 4] >= 0).all()


def test_bad_arguments_are_rejected(be):
    from nvrx_straggler import _native

    lib = _native.load()
    x = torch.zeros(16, device="cuda")
    c = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.zeros(8, device="cuda")
    assert lib.nvrx_row_stats(x.data_ptr(), c.data_ptr(), None, 1, 6, out.data_ptr(), None) == -22
    assert b"multiple of 4" in lib.nvrx_last_error()
    assert lib.nvrx_row_stats(None, c.data_ptr(), None, 1, 8, out.data_ptr(), None) == -22
    big = 65536 + 4
    assert lib.nvrx_row_stats(x.data_ptr(), c.data_ptr(), None, 1, big, out.data_ptr(), None) == -34
    ctx = ctypes.c_void_p()
    assert lib.nvrx_ctx_create(0, 1, 4, 70000, ctypes.byref(ctx)) == -34
    assert lib.nvrx_ctx_create(99, 1, 4, 64, ctypes.byref(ctx)) == -22


@pytest.mark.parametrize("cap,n", [(7, 21), (4, 3), (4, 4), (4, 5), (32, 100), (8192, 10000), (100, 1000)])
def test_ring_overwrite_oldest_matches_reference(be, cap, n):
    """Device ring == reference CircularBuffer / deque(maxlen): newest `cap` samples survive."""
    rings = be.make_rings(1, 2, cap)
    try:
        vals = (np.arange(n, dtype=np.float32) * 0.5 + 1.0)
        row = rings.row_for(0, "a")
        for v in vals[: n // 2]:
            rings.push(row, float(v))
        rings.push_many(row, vals[n // 2 :])
        assert rings.count(row) == min(n, cap)
        stored = rings.read_row(row)[: min(n, cap)]
        exp = oracle.ring_run(vals, cap)
        assert sorted(stored.tolist()) == sorted(exp.tolist())
        stats = rings.peek_stats()[row]
        e = oracle.section_stats(exp.astype(np.float64))
        assert stats[2] == np.float32(e[2]) and stats[5] == exp.size
        # device-side append continues the same ring
        extra = torch.arange(5, dtype=torch.float32, device="cuda") + 1000.0
        rings.push_device(row, extra)
        exp2 = oracle.ring_run(np.concatenate([vals, extra.cpu().numpy()]), cap)
        stored2 = rings.read_row(row)[: exp2.size]
        assert sorted(stored2.tolist()) == sorted(exp2.tolist())
        rings.reset()
        assert rings.count(row) == 0
        assert np.isnan(rings.peek_stats()[row][2])
    finally:
        rings.close()


@pytest.mark.parametrize("cap,n1,n2", [(64, 10, 20), (64, 50, 30), (64, 64, 64), (64, 5, 200), (8192, 10000, 100), (100, 0, 100)])
@pytest.mark.parametrize("together", [True, False])
def test_a_matrix_appended_to_consecutive_rows_at_once_matches_the_ring_of_the_reference(be, cap, n1, n2, together):
    """``nvrx_ring_push_device_rows``: a [rows][n] device matrix (also a strided view of a wider one) appended to consecutive
    rows in one call leaves every ring as the reference's deque(maxlen) / CircularBuffer would be (straggler.py:80-83,
    CircularBuffer.h:53-61) -- rows standing at one ring position (one strided copy per segment, wrapping included) and rows
    that do not (the row-by-row path)."""
    R = 5
    rings = be.make_rings(1, R + 2, cap)
    try:
        rows = [rings.row_for(0, f"s{r}") for r in range(R + 1)]
        assert rows == list(range(rows[0], rows[0] + R + 1))
        rng = np.random.default_rng(cap * 1000 + n1 + n2)
        a = rng.random((R, max(n1, 1)), dtype=np.float32)[:, :n1]
        wide = rng.random((R, n2 + 7), dtype=np.float32)
        hist = [[] for _ in range(R)]
        if n1:
            rings.push_device_rows(rows[0], torch.from_numpy(np.ascontiguousarray(a)).cuda())
            for r in range(R):
                hist[r] += a[r].tolist()
        if not together:                                  # row 2 moves ahead of the others
            rings.push(rows[2], 123.0)
            hist[2].append(123.0)
        view = torch.from_numpy(wide).cuda()[:, 3 : 3 + n2]   # leading dimension n2 + 7
        rings.push_device_rows(rows[0], view)
        for r in range(R):
            hist[r] += wide[r, 3 : 3 + n2].tolist()
        for r in range(R):
            exp = oracle.ring_run(np.asarray(hist[r], dtype=np.float32), cap)
            assert rings.count(rows[r]) == exp.size
            stored = rings.read_row(rows[r])[: exp.size]
            assert sorted(stored.tolist()) == sorted(exp.tolist()), r
        assert rings.count(rows[R]) == 0                 # the neighbour row was not touched
        stats = rings.peek_stats()
        for r in range(R):
            exp = oracle.ring_run(np.asarray(hist[r], dtype=np.float32), cap)
            e = oracle.section_stats(exp.astype(np.float64))
            assert stats[rows[r]][2] == np.float32(e[2]) and stats[rows[r]][5] == exp.size
    finally:
        rings.close()


def test_a_matrix_append_that_wraps_in_the_last_rows_of_the_ring_storage_stays_inside_it(be):
    """The strided copy of ``nvrx_ring_push_device_rows`` starts in the middle of a row when the rings are part full: for the
    LAST rows of the ring storage the pitch-times-height span of that copy reaches past the allocation (only the span, not
    the bytes copied).  The runtime must neither refuse nor overrun: every ring, the last one included, holds exactly the
    newest samples, again and again."""
    cap, R = 96, 6
    rings = be.make_rings(1, R, cap)              # the matrix covers ALL rows: the last row ends the allocation
    try:
        rows = [rings.row_for(0, f"s{r}") for r in range(R)]
        assert rows == list(range(R))
        rng = np.random.default_rng(3)
        hist = [[] for _ in range(R)]
        for it in range(40):
            n = int(rng.integers(1, 2 * cap))
            m = rng.random((R, n), dtype=np.float32)
            rings.push_device_rows(0, torch.from_numpy(m).cuda())
            for r in range(R):
                hist[r] += m[r].tolist()
        torch.cuda.synchronize()
        for r in range(R):
            exp = oracle.ring_run(np.asarray(hist[r], dtype=np.float32), cap)
            stored = rings.read_row(rows[r])[: exp.size]
            assert sorted(stored.tolist()) == sorted(exp.tolist()), r
    finally:
        rings.close()


@pytest.mark.parametrize("cap,rows,n", [(16, 40, 3000), (64, 300, 20000), (100, 4096, 409600)])
def test_bulk_append_of_row_value_pairs_matches_the_ring_of_the_reference(be, cap, rows, n):
    """nvrx_ring_push_pairs (the per-kernel tracer's route into the rings: one scatter launch for all keys) == the
    reference's CircularBuffer fed pair by pair (CuptiProfiler.cpp:186-207, CircularBuffer.h:53-61): rows that receive
    more than a ring's worth in ONE call, rows continued by a second call and by single pushes, skipped (negative) rows.
    The largest case is the reference's own data_shared sizing: 4096 kernel keys x 100 samples (test_data_shared.py:62-66)."""
    rng = np.random.default_rng(cap)
    rings = be.make_rings(1, rows, cap)
    try:
        for r in range(rows):
            rings.row_for(1, f"k{r}")
        # skewed: a few rows take most of the pairs (they wrap inside one call), many rows take a handful
        hot = rng.integers(0, min(rows, 8), n // 2)
        cold = rng.integers(0, rows, n - n // 2)
        row_of = rng.permutation(np.concatenate([hot, cold])).astype(np.int32)
        row_of[rng.integers(0, n, n // 50)] = -1
        vals = rng.normal(100.0, 5.0, n).astype(np.float32)
        a = n * 2 // 3
        rings.push_pairs(row_of[:a], vals[:a])
        for i in range(a, min(a + 50, n)):           # single pushes in between keep their place in the order
            if row_of[i] >= 0:
                rings.push(int(row_of[i]), float(vals[i]))
        rings.push_pairs(row_of[min(a + 50, n):], vals[min(a + 50, n):])
        stats = rings.peek_stats()
        check = range(rows) if rows <= 300 else list(range(8)) + rng.integers(0, rows, 120).tolist()
        for r in check:
            mine = vals[row_of == r]
            exp = oracle.ring_run(mine, cap)
            assert rings.count(r) == exp.size, r
            stored = rings.read_row(r)[: exp.size]
            assert sorted(stored.tolist()) == sorted(exp.tolist()), r
            if exp.size:
                e = oracle.kernel_stats(exp)
                assert stats[r][0] == e[0] and stats[r][1] == e[1] and stats[r][5] == exp.size, r
    finally:
        rings.close()


def test_tracer_records_reach_their_rows_through_the_native_sink(be):
    """The per-kernel tracer's path into the device rings (``nvrx_ktrace_feed`` -> key cache -> ``nvrx_ring_push_staged``;
    what the rocprofiler-sdk callback thread does with its batches) on synthetic dispatches: 4096 kernel keys x 100
    durations in batches of ~1000; every key's statistics row equals the oracle's computeStats of its own durations."""
    from nvrx_straggler import ktrace

    K, per = 4096, 100
    rng = np.random.default_rng(7)
    base = 1 << 44
    for k in range(K):
        ktrace.feed_kernel_name(base + k, f"k{k:04d}")
    d = np.zeros(K * per, dtype=ktrace.DISPATCH_DTYPE)
    kid = rng.permutation(np.repeat(np.arange(K, dtype=np.uint64), per))
    d["kernel_id"] = base + kid
    d["workgroup"], d["grid"], d["start_ns"] = (64, 1, 1), (64 * 3, 1, 1), 1000
    ns = (rng.lognormal(3.0, 0.4, K * per) * 1000.0).astype(np.uint64) + 1
    d["end_ns"] = 1000 + ns
    us = ns.astype(np.float32) / np.float32(1000.0)
    ktrace.KernelTraceProfiler._live = None
    prof = ktrace.KernelTraceProfiler(statsMaxLenPerKernel=128, max_keys=K)
    try:
        for lo in range(0, d.size, 1000):
            ktrace.feed(d[lo:lo + 1000])
        assert prof.harvest(wait=True) == 0
        stats = prof._rings.peek_stats()
        rows = prof._rings.kernel_row_names
        assert len(rows) == K and prof.keys_without_row == 0
        for k in rng.integers(0, K, 200).tolist():
            e = oracle.kernel_stats(us[kid == k])
            row = rows[f"k{k:04d}_blk_64_1_1_grid_3_1_1"]
            assert stats[row][5] == per and stats[row][0] == e[0] and stats[row][1] == e[1] and stats[row][2] == e[2], k
    finally:
        prof.close()
        ktrace.KernelTraceProfiler._live = None


@pytest.mark.parametrize("stride", [1024, 4096, 10000])
def test_non_finite_samples_terminate_and_leave_the_other_rows_exact(be, stride):
    """Durations are finite by construction (clock differences), but ``cpu_elapsed_times.extend`` takes whatever a caller hands
    it.  A row holding +inf / -inf / NaN must not hang the selection (keys are raw bit patterns: a total order whatever the
    payload) nor disturb its neighbours; its own MIN / MAX / MED are the reference's wherever the reference's are defined
    (infinities order like numbers in torch.min / max / median; a NaN makes the reference's answers NaN -- here only "finite
    rows stay exact and the launch returns" is pinned for it)."""
    rng = np.random.default_rng(stride)
    rows = 12
    m = rng.lognormal(1.0, 0.4, (rows, stride)).astype(np.float32)
    counts = np.full(rows, stride, dtype=np.uint32)
    kinds = np.array([0, 1] * (rows // 2), dtype=np.uint8)
    m[2, 5] = np.inf
    m[3, 7] = -np.inf
    m[4, : stride // 2 + 3] = np.inf          # the median itself is +inf
    m[5, 11] = np.nan
    m[6, ::3] = np.nan
    m[7, :] = np.inf
    got = _run(be, m, counts, kinds)          # (a hang would be the test's timeout)
    assert got.shape[0] == rows and (got[:, 5] == stride).all()
    finite_rows = [0, 1, 8, 9, 10, 11]
    _check_against_oracle(got[finite_rows], m[finite_rows], counts[finite_rows], kinds[finite_rows], "finite rows next to non-finite ones")
    # infinities: selections as numbers
    assert got[2, 1] == np.inf and got[2, 0] == m[2][np.isfinite(m[2])].min()
    assert got[2, 2] == np.float32(oracle.rows_stats(m[2:3], counts[2:3], kinds[2:3])[0, 2])
    assert got[3, 0] == -np.inf and got[3, 2] == np.float32(oracle.rows_stats(m[3:4], counts[3:4], kinds[3:4])[0, 2])
    assert got[4, 2] == np.inf and got[7, 0] == np.inf and got[7, 1] == np.inf and got[7, 2] == np.inf
