"""Model of an ATOMICS-FREE Resample2d backward, to be read BEFORE anything is built.

Decomposition.  Pass 1 = today's kernel up to the flush (LDS windows of tile +- 16 px accumulate the scatter, grad_flow is gathered), but
a tile's window leaves as plain 16-byte row stores into ITS OWN halo plane (64 x 96 x 3 floats per tile; all-zero 64-byte segments are
skipped and recorded in a per-tile bitmap) instead of global atomics onto the image gradient; far pixels (corners outside the window) go
to per-destination-tile lists.  Pass 2 = one workgroup per 32 x 64 output tile sums the <= 9 planes that cover it, applies its far list
and OVERWRITES grad_input1 (no zero fill).

Measured inputs (profiles/):
  r05_c_resample_timeline.log   the kernel WITHOUT its flush ends at 30.8 us (loads 3.2, LDS compare-and-swap scatter 9.4 + 8.1 over the
                                two rounds of workgroups, gather, ramp); with the atomic flush at 49.9 us
  r05_a_atomic_rate.log         fp32 atomics: 20.5 G requests/s; plain stores of the same 16-lane pattern: 29.3 G groups/s
  bench.py (r05/r06)            torch's zero fill of 18.9 MB: 5.7 us (3.3 TB/s of pure writes); device copy 5.45 TB/s (read + write)
                                a launch boundary between two dependent kernels: 2-3 us

The counts come from the bench's flow through resample_atomic_requests.count (same tiles, same windows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

WRITE_TBPS = 3.3          # pure-write stream (the fill's rate)
MIXED_TBPS = 4.5          # 2/3 reads + 1/3 writes (between the fill and the copy ceiling)
NO_FLUSH_US = 30.8        # today's kernel with the flush removed
GAP_US = 2.5              # launch boundary between dependent kernels
TODAY_US = 59.9           # event pair around zero fill + kernel (51.1 us kernel alone)


def model(counts, B=8, C=3, H=384, W=512):
    seg_bytes = 64
    flush_segments = counts["flush_req"] * B                 # non-zero 64-byte segments of all windows, all channels
    plane_bytes = flush_segments * seg_bytes
    far_entries = counts["far_px"] * B * 4                   # one 16-byte entry (pixel index, 3 channels) per corner
    image_bytes = B * C * H * W * 4
    # pass 1: everything but the flush as today; the flush becomes a write stream that overlaps the other workgroups' scatter only
    # partly (it starts when the first windows are complete, 9-17 us in): charge half of it
    flush_us = plane_bytes / (WRITE_TBPS * 1e6)
    pass1_us = NO_FLUSH_US + 0.5 * flush_us
    # pass 2: read the non-zero segments + far lists + bitmaps, write the image; one ramp (2 us) of its own
    pass2_bytes = plane_bytes + far_entries * 16 + image_bytes
    pass2_us = 2.0 + pass2_bytes / (MIXED_TBPS * 1e6)
    total = pass1_us + GAP_US + pass2_us
    return dict(plane_MB=plane_bytes / 1e6, far_entries=far_entries, flush_us=flush_us, pass1_us=pass1_us, pass2_us=pass2_us,
                total_us=total, today_us=TODAY_US, workspace_MB=B * (H // 32) * (W // 64) * 64 * 96 * C * 4 / 1e6)


if __name__ == "__main__":
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "resample_atomic_requests.py")).read().split("for shape in (")[0]
    ns = {}
    exec(compile(src, "resample_atomic_requests.py", "exec"), ns)
    m = model(ns["count"](32, 64, 16, b=0))
    print({k: round(v, 1) for k, v in m.items()})
    print("verdict: %.1f us all-in against %.1f us today (%.0f %% of it); the bar for building it was <= 45 us -> %s" %
          (m["total_us"], m["today_us"], 100 * m["total_us"] / m["today_us"], "build" if m["total_us"] <= 45 else "NOT built"))
