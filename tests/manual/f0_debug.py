import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from audio_feature_extraction_amd import _native as N
from oracle import pyin_ref as P
SR = 22050
def voiced_tone(freq, seconds, vib=0.0, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(int(SR * seconds)) / SR
    f = freq * (1 + vib * np.sin(2 * np.pi * 5 * t))
    ph = 2 * np.pi * np.cumsum(f) / SR
    y = 0.3 * np.sin(ph) + 0.1 * np.sin(2 * ph + 0.3) + 0.05 * np.sin(3 * ph + 1.0)
    y += 0.005 * rng.standard_normal(t.size)
    return y.astype(np.float32)
rng = np.random.default_rng(11)
clips = []; kinds = []
for i in range(8):
    parts = []; kk = []
    for _ in range(int(rng.integers(2, 5))):
        n = int(rng.integers(1500, 7000))
        kind = rng.integers(0, 4)
        if kind == 0: parts.append(np.zeros(n, np.float32))
        elif kind == 1: parts.append((0.05 * rng.standard_normal(n)).astype(np.float32))
        else:
            f = float(rng.uniform(80, 900))
            seg = voiced_tone(f, n / SR, vib=float(rng.uniform(0, 0.03)), seed=int(rng.integers(1 << 30)))[:n]
            parts.append(float(rng.uniform(0.1, 1.0)) * seg)
        kk.append((int(kind), n))
    clips.append(np.concatenate(parts).astype(np.float32)); kinds.append(kk)
ctx = N.Context(0); plan = N.Plan(ctx, N.make_params(SR, 1024, 256, 13))
for i, c in enumerate(clips):
    os.environ["AFX_F0_DUMP"] = "/tmp/f0dump.bin"
    out = plan.f0_batch(c, np.zeros(1, np.int64), np.array([c.size], np.int64), P.C2_HZ, P.C7_HZ, flags=0, want_frames=True)
    T = 1 + c.size // 256
    g = out["f0_flat"][:T]
    r, vf, vp, inter = P.pyin(c, sr=SR, frame_length=1024, hop_length=256, return_internal=True)
    same = np.isnan(g) == np.isnan(r); v = ~np.isnan(g) & ~np.isnan(r); same[v] &= np.abs(g[v] - r[v]) <= 1e-9 * r[v]
    print("clip", i, kinds[i], "T", T, "match", same.mean())
    bad = np.flatnonzero(~same)
    for t in bad[:12]:
        obs = inter["obs"][:601, t]
        nz = np.flatnonzero(obs)
        print("   t", t, "gpu", g[t], "ref", r[t], "vp", vp[t], "ncand", nz.size, "top", nz[np.argsort(-obs[nz])][:3], np.sort(obs[nz])[::-1][:3])

    raw = open("/tmp/f0dump.bin", "rb").read()
    frames, cap = np.frombuffer(raw[:16], np.int64)
    o = 16
    cnt = np.frombuffer(raw[o:o + 4 * frames], np.int32); o += 4 * frames
    gvp = np.frombuffer(raw[o:o + 8 * frames], np.float64); o += 8 * frames
    bn = np.frombuffer(raw[o:o + 2 * frames * cap], np.int16).reshape(frames, cap); o += 2 * frames * cap
    pr = np.frombuffer(raw[o:o + 8 * frames * cap], np.float64).reshape(frames, cap)
    worst = 0
    for t in range(T):
        obs = inter["obs"][:601, t]
        g = np.zeros(601)
        for j in range(cnt[t]):
            if 0 <= bn[t, j] < 601: g[bn[t, j]] = pr[t, j]
        d = np.abs(g - obs).max()
        if d > 1e-12 or abs(gvp[t] - vp[t]) > 1e-12:
            print("   obs differ t", t, "max abs", d, "vp gpu/ref", gvp[t], vp[t], "cnt", cnt[t], "nz ref", np.count_nonzero(obs),
                  "gpu bins", bn[t, :cnt[t]][:12], "ref bins", np.flatnonzero(obs)[::-1][:12])
    exact = sum(1 for t in range(T) if gvp[t] == vp[t])
    one = [(float(1 - gvp[t]), float(1 - vp[t])) for t in range(T) if (gvp[t] != vp[t])]
    print("   voiced_prob bit-identical in", exact, "of", T, "frames; differing (1-vp) pairs:", one[:6])
