import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import workloads as W
lat, lon = W.columns_from_mask("N145")
w = W.make_workload("richards", lat, lon, 32)
d = W.setup_device(w)
d.step(w["dt"], 5, False)   # warm
d.step(w["dt"], 1, False)
g = d.get("tend_internal_energy")           # [Nz][Nh] view of the device buffer [Nh][Nzp]
buf = np.ascontiguousarray(g.T).ravel()       # back to device order: column-major in z (Nzp = 32 = Nz here)
nw = (lat.size + 1) // 2
st = buf[: nw * 4].reshape(nw, 4)
t0, dur, hw = st[:, 0], st[:, 1], st[:, 2].astype(np.uint64)
t0 = t0 - t0.min()
end = t0 + dur
print("waves", nw, "; s_memtime ticks (100 MHz => 10 ns)")
print("kernel span:", end.max(), " wave lifetime min/med/p90/max:", dur.min(), np.median(dur), np.percentile(dur, 90), dur.max())
print("mean waves in flight:", dur.sum() / end.max(), " per SIMD:", dur.sum() / end.max() / 1024)
wave_id = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("wave_id histogram:", np.bincount(wave_id.astype(int)))
print("distinct (se,sh,cu):", len(set(zip(se.tolist(), sh.tolist(), cu.tolist()))), " simd hist:", np.bincount(simd.astype(int)))
xcc = (st[:, 3].astype(np.uint64) & 0xf).astype(int)
print("xcc histogram:", np.bincount(xcc))
t0raw = st[:, 0]
cuid = (((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)).astype(int)
key = xcc * 1024 + cuid
spans, meanfl = [], []
for n, kk in enumerate(sorted(set(key.tolist()))):
    m = key == kk
    a = t0raw[m] - t0raw[m].min(); e = a + dur[m]
    span = e.max(); spans.append(span); meanfl.append(dur[m].sum() / span)
    if n in (0, 100, 200):
        T = np.linspace(0, span, 40)
        print(f"CU {kk}: waves {m.sum()} span {span:.0f} ticks, mean in flight {dur[m].sum() / span:.1f}")
        print("  in flight:", [int(np.sum((a <= t) & (t < e))) for t in T])
        print("  starts/bin:", np.histogram(a, bins=39, range=(0, span))[0].tolist())
        o = np.argsort(a)
        print("  first 40 start times:", a[o][:40].astype(int).tolist())
        print("  durations in start order (every 4th):", dur[m][o][::4].astype(int).tolist())
spans = np.array(spans); meanfl = np.array(meanfl)
print("CUs:", len(spans), " span min/med/max:", spans.min(), np.median(spans), spans.max(), " mean in flight min/med/max:", meanfl.min(), np.median(meanfl), meanfl.max())
