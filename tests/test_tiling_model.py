"""Host-side model of the correlation kernels' tilings (no GPU): the wave-role / block-pair tables the HIP sources hold as
constexpr functions, restated here and checked for what the kernels rely on -- every (centre block, neighbour block) pair inside
the displacement band is computed by exactly one wave, in exactly one pass, and the per-wave counts are the array sizes the
kernels declare.  The constants are also read back from the sources, so that an edit there without an edit here fails."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "flownet2-pytorch_amd", "csrc")


def src(name):
    with open(os.path.join(CSRC, name)) as f:
        return f.read()


# ---- forward, maps up to 64 px: 8 A blocks x 8 B blocks per parity (csrc/f16x2_common.h)
FWD_ROLES = {0: (0, 3), 1: (1, 2), 2: (4, 7), 3: (5, 6)}
FWD_M = {0: (0, 6), 1: (0, 5), 2: (1, 7), 3: (2, 7)}


def test_forward_roles_cover_every_block_pair_once():
    seen = {}
    for role, blocks in FWD_ROLES.items():
        lo, hi = FWD_M[role]
        pairs = [(a, m) for m in range(lo, hi + 1) for a in blocks if abs(a - m) <= 3]
        assert len(pairs) == 11, (role, len(pairs))                       # NP
        for p in pairs:
            assert p not in seen, p
            seen[p] = role
        for a in blocks:                                                 # the role's B range holds every block its A blocks meet
            assert all(lo <= m <= hi for m in range(max(0, a - 3), min(7, a + 3) + 1))
    want = {(a, m) for a in range(8) for m in range(8) if abs(a - m) <= 3}
    assert set(seen) == want and len(want) == 44
    s = src("f16x2_common.h")
    assert "constexpr int NP = 11;" in s and "constexpr int DR = 10, D = 2 * DR + 1, NU = 6;" in s
    assert "role == 0 ? (ab ? 3 : 0) : role == 1 ? (ab ? 2 : 1) : role == 2 ? (ab ? 7 : 4) : (ab ? 6 : 5)" in s


def test_forward_row_blocks_cover_the_displacement_rows_once():
    """Task (rg, u): A rows 4rg .. 4rg+3 against B rows 4rg - 10 + 4u .. +3; plane (ai, bi) holds displacement row tj = 4u + bi - ai.
    Over u = 0..5 every tj in 0..20 appears exactly once for each A row ai."""
    for ai in range(4):
        tjs = [4 * u + bi - ai for u in range(6) for bi in range(4)]
        inside = [t for t in tjs if 0 <= t < 21]
        assert sorted(inside) == list(range(21)), (ai, sorted(inside))


# ---- forward, wider maps: 4 A' blocks x the 10 B' blocks they meet (csrc/correlation_f16x2_wide.hip)
def test_wide_forward_windows():
    s = src("correlation_f16x2_wide.hip")
    assert "constexpr int AW = 4;" in s and "constexpr int NB = 7;" in s
    for W in (72, 96, 128, 200, 256):
        nblk = W // 8
        nxq = (W + 31) // 32
        seen = set()
        for xq in range(nxq):
            for a in range(4):                       # the wave's A' block
                A = 4 * xq + a
                for jj in range(7):                  # B' block a + jj of the window = image block 4 xq - 3 + a + jj
                    m = 4 * xq - 3 + a + jj
                    slot = 4 + a + jj                # LDS block slot: 0..3 A', 4..15 B'
                    assert 4 <= slot <= 13
                    assert m - A == jj - 3
                    if A < nblk and 0 <= m < nblk:
                        assert (A, m) not in seen
                        seen.add((A, m))
        want = {(A, m) for A in range(nblk) for m in range(nblk) if abs(A - m) <= 3}
        assert seen == want, W


# ---- backward: neighbour block PAIRS (2j, 2j+1) per centre block
BWD_ROLES = FWD_ROLES


def bwd_meets(a, j, off=0):
    return off + 2 * j + 1 >= a - 3 and off + 2 * j <= a + 3


def test_backward_roles():
    for role, blocks in BWD_ROLES.items():
        n = sum(bwd_meets(a, j) for a in blocks for j in range(4))
        assert n == 6, (role, n)                                          # NF
    for a in range(8):                                                   # every neighbour block in the band lies in a pair that is met
        for m in range(max(0, a - 3), min(7, a + 3) + 1):
            assert bwd_meets(a, m // 2)
    for s_name in ("correlation_f16x2_bwd.hip", "correlation_f16_bwd.hip"):
        assert "constexpr int NF = 6;" in src(s_name)


def test_wide_backward_passes():
    """Centre window of 8 blocks; pass 0 walks the neighbour blocks -3 .. 4 of the window, pass 1 the blocks 5 .. 12; the centre
    blocks of a wave are (r, 7 - r)."""
    s = src("correlation_f16x2_bwd_wide.hip")
    assert "return ps ? 5 : -3;" in s and "return ab ? 7 - role : role;" in s and "constexpr int NF = 5;" in s
    PO = (-3, 5)
    for role in range(4):
        blocks = (role, 7 - role)
        counts = [sum(bwd_meets(a, j, PO[ps]) for a in blocks for j in range(4)) for ps in range(2)]
        assert counts == [5, 3], (role, counts)
    for a in range(8):
        covered = {}
        for ps in range(2):
            for j in range(4):
                if bwd_meets(a, j, PO[ps]):
                    for blk in range(2):
                        m = PO[ps] + 2 * j + blk
                        if abs(m - a) <= 3:
                            assert m not in covered, (a, m)
                            covered[m] = ps
        assert sorted(covered) == list(range(a - 3, a + 4)), (a, sorted(covered))
    # pass 1 is skipped exactly when none of its blocks that a centre block of the image needs lies inside the image
    for W in (72, 80, 104, 128, 136, 200):
        nblk = W // 8
        for xw in range((W + 63) // 64):
            need = any(8 * xw + a < nblk and 8 * xw + 5 <= m < nblk and abs(m - (8 * xw + a)) <= 3
                       for a in range(8) for m in range(8 * xw + 5, 8 * xw + 13))
            runs = 64 * xw + 8 * 5 < W
            assert runs == need, (W, xw)


def test_gather_band_constants():
    """The gather reads slot (bis, bjs) of neighbour block dm = dj + blk at displacement column ti = 10 +- (4 dm + bjs - aj); the kernels
    test `vs = 4 blk - aj` against per-slot constants instead of ti itself: same predicate."""
    for dj in range(-4, 4):
        for bjs in range(4):
            hi, lo = 10 - 4 * dj - bjs, -10 - 4 * dj - bjs
            for blk in range(2):
                for aj in range(4):
                    vs = 4 * blk - aj
                    t = 4 * (dj + blk) + bjs - aj
                    assert (lo <= vs <= hi) == (-10 <= t <= 10)
            check = dj < -1 or dj > 0
            always = all(-10 <= 4 * (dj + blk) + bjs - aj <= 10 for blk in range(2) for aj in range(4) for bjs in range(4))
            assert check == (not always), dj


def test_task_table_limits_in_sources():
    s = src("f16x2_common.h")
    m = re.search(r"constexpr int MAX_TAB = (\d+);", s)
    assert m and int(m.group(1)) == 768            # 2 parities x NRG x 6 row blocks <= 768  ->  H <= 512
    assert 2 * ((512 // 2 + 3) // 4) * 6 == 768


def test_resample_backward_atomic_request_model():
    """DESIGN.md 4.6's cost model of the Resample2d backward (scripts/design/resample_atomic_requests.py): global fp32 atomics cost per
    REQUEST = per aligned 64-byte segment an instruction touches.  Brute force on a small field: the model's count of flush segments equals
    the number of distinct (tile, plane row, 16-float segment) triples that receive a non-zero contribution, and for the bench's flow the
    model gives the 0.59 M + 0.20 M requests the document quotes."""
    import importlib.util
    import os
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("req", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "design",
                                                                       "resample_atomic_requests.py"))
    src = open(spec.origin).read().split("for shape in (")[0]          # the definitions, not the table it prints
    ns = {}
    exec(compile(src, spec.origin, "exec"), ns)
    r = ns["count"](32, 64, 16, b=0)
    flush, far_now, far_px = r["flush_req"] * 8, r["far_req_now"] * 8, r["far_px"] * 8
    assert 0.55e6 < flush < 0.63e6 and 0.18e6 < far_now < 0.23e6 and 16e3 < far_px < 21e3
    assert r["far_req_merged"] < 0.6 * r["far_req_now"]                 # two corners of a row in one request: about half
    # at 20.5 G requests/s: 36-42 us of atomic-unit time
    assert 36.0 < (flush + far_now) / 20.5e3 < 42.0


def test_resample_backward_halo_plane_model():
    """the atomics-free decomposition of the Resample2d backward (per-tile halo planes written with plain stores + one
    accumulate pass that overwrites grad_input1) is MODELLED before anything is built (scripts/design/resample_halo_planes.py): with the
    measured no-flush time of today's kernel (30.8 us: the LDS compare-and-swap scatter is the floor either way), the plane traffic of
    the bench's flow (38 MB of non-zero 64-byte segments each way) and the measured stream rates it comes out at 50-58 us against 59.9
    today -- not the <= 45 us that would justify a second kernel, a 57 MB workspace and far-pixel lists.  Not built."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "design")
    spec = importlib.util.spec_from_file_location("halo", os.path.join(here, "resample_halo_planes.py"))
    halo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(halo)
    ns = {}
    exec(compile(open(os.path.join(here, "resample_atomic_requests.py")).read().split("for shape in (")[0], "req", "exec"), ns)
    counts = ns["count"](32, 64, 16, b=0)
    m = halo.model(counts)
    assert 35.0 < m["plane_MB"] < 41.0 and 60e3 < m["far_entries"] < 85e3 and 55.0 < m["workspace_MB"] < 58.0
    assert 50.0 < m["total_us"] < 58.0 and m["total_us"] > 45.0          # the decision: above the bar
    assert m["pass1_us"] >= halo.NO_FLUSH_US                             # the scatter floor is in both designs
    # brute force of the plane traffic on a small field: non-zero segments = distinct (tile, row, segment) triples, as in the atomic model
    assert counts["flush_req"] * 8 * 64 == int(round(m["plane_MB"] * 1e6))
