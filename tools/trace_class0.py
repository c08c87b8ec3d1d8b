"""Summarise the class-0 (producer / consumer wave pairs) records of a render trace written with
GRPG_RENDER_TRACE by the `trace` / `only0trace` experiment builds (tools/trace_render.py): per quarter the
life time of its consumer and producer wave and the share of it each spent waiting for the other."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint32)
nw = raw.size // 17
t = raw[:nw * 8].reshape(-1, 8)
sel = (t[:, 0] != 0xFFFFFFFF) & ((t[:, 0] >> 30) == 1)
v = t[sel]
if not len(v):
    print("no class-0 records"); sys.exit(0)
idx = np.nonzero(sel)[0]
block = idx // 4
wave = v[:, 7] & 15
role = wave >> 1                     # waves 0, 1: consumers of the workgroup's two quarters; 2, 3: producers
block = block * 2 + (wave & 1)       # one id per quarter
wait_cyc = (v[:, 7] >> 4).astype(np.int64) * 256          # shader clocks
us = v[:, 5].astype(np.int64) / 100.0                       # wall clock: 100 MHz
names = {0: "consumer", 1: "producer"}
print("class-0 quarters: %d, tiles %d" % (len(np.unique(block)), len(np.unique(v[:, 0]))))
for r in range(2):
    m = role == r
    if not m.any():
        continue
    w_us = wait_cyc[m] / 2000.0      # ~2 GHz under load
    print("%-10s life us: mean %.1f max %.1f | waiting (est. us at 2 GHz): mean %.1f, share of life %.2f | "
          "batches/chunks mean %.1f, survivors mean %.0f, waits mean %.1f" % (
              names[r], us[m].mean(), us[m].max(), w_us.mean(), (w_us / np.maximum(us[m], 1e-9)).mean(),
              v[m, 2].mean(), v[m, 3].mean(), v[m, 4].mean()))
# the ten longest quarters
bl = v[role == 0]
blk = block[role == 0]
order = np.argsort(-bl[:, 5].astype(np.int64))[:10]
for i in order:
    b = blk[i]
    parts = []
    for r in range(2):
        m = (block == b) & (role == r)
        if m.any():
            j = np.nonzero(m)[0][0]
            parts.append("%s %.1f us (wait %.1f)" % (names[r][:4], us[j], wait_cyc[j] / 2000.0))
    print("  tile %d q%d len %d batches %d survivors %d: %s" % (bl[i, 0] & 0x3FFFFFFF, b & 3, bl[i, 1], bl[i, 2], bl[i, 3],
                                                              "; ".join(parts)))
