"""Summarise a render trace written with GRPG_RENDER_TRACE (see tools/trace_render.py)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint32)
nw = raw.size // 17
t = raw[:nw * 8].reshape(-1, 8)
tloop = raw[nw * 8:nw * 9]
bucket = raw[nw * 9:].reshape(-1, 8)   # BLEND iterations / cycles by live-pixel count (<=2, <=8, <=24, >24)
# (records of the class-0 producer / consumer waves carry bit 30 in their tile word: tools/trace_class0.py)
sel = (t[:, 0] != 0xFFFFFFFF) & (((t[:, 0] >> 30) & 1) == 0)
bucket = bucket[sel].astype(np.int64)
v = t[sel]
tloop = tloop[sel].astype(np.int64)
tstage = (v[:, 7] >> 4).astype(np.int64) * 256
tile = v[:, 0] & 0x7FFFFFFF; light = (v[:, 0] >> 31) == 1
b = v[:, 2]; sv = v[:, 3]; bl = v[:, 4]; cyc = v[:, 5].astype(np.int64); st = v[:, 6].astype(np.int64)
end = (st - st.min() + cyc) & 0xFFFFFFFF
print("kernel span us %.1f" % (end.max() / 100))
for name, m in (("heavy", ~light), ("light", light)):
    print("%s waves %d wave-time sum %.1f ms batches %d survivors %d blend-iters %d  us/batch %.3f" % (
        name, m.sum(), cyc[m].sum() / 1e5, b[m].sum(), sv[m].sum(), bl[m].sum(), cyc[m].sum() / max(b[m].sum(), 1) / 100))
for i in np.argsort(-cyc)[:3]:
    print("  tile %d wave %d len %d batches %d surv %d blends %d  %.1f us  %.3f us/batch  stage %d cyc/batch  loop %d cyc/batch" % (
        tile[i], v[i, 7] & 15, v[i, 1], b[i], sv[i], bl[i], cyc[i] / 100, cyc[i] / max(b[i], 1) / 100,
        tstage[i] / max(b[i], 1), tloop[i] / max(b[i], 1)))
print("heavy BLEND iterations by live pixels of the quarter (<=2, <=8, <=24, >24):")
hb = bucket[~light]
print("  all heavy waves: iters", hb[:, :4].sum(0), " Mcycles", (hb[:, 4:].sum(0) / 1e6).round(1))
for i in np.argsort(-cyc)[:3]:
    print("  tile %d wave %d: iters %s kcycles %s" % (tile[i], v[i, 7] & 15, bucket[i, :4], (bucket[i, 4:] / 1e3).round(0)))
for name, m in (("heavy", ~light), ("light", light)):   # cull + compaction vs the blend loop, shader cycles
    tot = max(tstage[m].sum() + tloop[m].sum(), 1)
    print("%s: cull + compaction %.1f M cycles (%.2f), blend loop %.1f M cycles (%.2f)" % (
        name, tstage[m].sum() / 1e6, tstage[m].sum() / tot, tloop[m].sum() / 1e6, tloop[m].sum() / tot))
