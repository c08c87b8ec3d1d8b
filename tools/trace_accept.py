import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint32)
nw = raw.size // 17
t = raw[:nw * 8].reshape(-1, 8)
bucket = raw[nw * 9:].reshape(-1, 8)
sel = t[:, 0] != 0xFFFFFFFF
light = (t[:, 0] >> 31) == 1
m = sel & ~light
b = bucket[m].astype(np.int64)
sv = t[m][:, 3].astype(np.int64).sum()
print("heavy survivors", sv, "accepting lane sum", b[:,0].sum(), "splats with >=1 accepting lane", b[:,1].sum(),
      "avg lanes per survivor %.2f" % (b[:,0].sum()/sv), "avg lanes per touching splat %.2f" % (b[:,0].sum()/max(b[:,1].sum(),1)))
print("left-half touched", b[:,2].sum(), "right-half touched", b[:,3].sum(), "sum/survivors %.3f" % ((b[:,2].sum()+b[:,3].sum())/sv),
      " touching-splats fraction %.3f" % (b[:,1].sum()/sv))
print("4x(4x4) groups touched per survivor %.3f ; 2x(16x2) row-pair groups %.3f" % (b[:,4].sum()/sv, b[:,5].sum()/sv))
