"""Driver of pq_accuracy.hip: random and 10-bit-like inputs, torch's CPU fp32 pq2lin as the yardstick."""
import subprocess, sys, os
import numpy as np, torch
rng = np.random.default_rng(0)
v = np.concatenate([rng.random(1 << 22), (rng.integers(0, 1024, 1 << 20) / 1023.0) * 0.9 + rng.random(1 << 20) * 1e-3]).astype(np.float32)
v.tofile("/tmp/pq_in.f32")
subprocess.check_call(["/tmp/pq_accuracy", "/tmp/pq_in.f32", "/tmp/pq0.f32", "/tmp/pq1.f32", "/tmp/pq2.f32", "/tmp/pq3.f32", "/tmp/pq4.f32", "/tmp/pq5.f32", "/tmp/pq6.f32", "/tmp/pq7.f32", "/tmp/pq8.f32", "/tmp/pq9.f32", "/tmp/pq10.f32"])
V = torch.tensor(v)
n, m, c1, c2, c3 = 0.15930175781250000, 78.843750000000000, 0.83593750000000000, 18.851562500000000, 18.687500000000000
im_t = torch.pow(V, 1 / m)
ref = (10000 * torch.pow((im_t - c1).clamp(min=0) / (c2 - c3 * im_t), 1 / n)).numpy()
V64 = V.double()
t64 = torch.pow(V64, 1 / m)
exact = (10000 * torch.pow((t64 - c1).clamp(min=0) / (c2 - c3 * t64), 1 / n)).numpy()
sel = exact > 0.005                           # what survives the display model's clip
print("torch fp32 vs exact (fp64): rel err  mean %.2e  99.9%% %.2e  max %.2e" % tuple(np.percentile(np.abs(ref - exact)[sel] / exact[sel], q) if q else (np.abs(ref - exact)[sel] / exact[sel]).mean() for q in (0, 99.9, 100)))
for k in (0, 1, 2, 5, 6, 7, 10):
    got = np.fromfile("/tmp/pq%d.f32" % k, np.float32)
    e = np.abs(got - ref)[sel] / ref[sel]
    same = (got == ref)[sel].mean()
    print("mode %d vs torch fp32: rel err  mean %.2e  99.9%% %.2e  max %.2e   identical %.1f %%" % (k, e.mean(), np.percentile(e, 99.9), e.max(), 100 * same))

t_cr = t64.float().numpy()                       # correctly rounded V^(1/m)
t_torch = im_t.numpy()
t_f64 = np.fromfile("/tmp/pq3.f32", np.float32)
t_fast = np.fromfile("/tmp/pq4.f32", np.float32)
ulp = np.spacing(t_cr)
for name, t in (("torch.pow fp32", t_torch), ("fp64 path", t_f64), ("hardware log2/exp2", t_fast)):
    d = (t.astype(np.float64) - t_cr.astype(np.float64)) / ulp
    print("t = V^(1/m), %-20s vs correctly rounded: identical %.1f %%, within 1 ulp %.1f %%, max %.1f ulp" % (name, 100 * (d == 0).mean(), 100 * (np.abs(d) <= 1).mean(), np.abs(d).max()))
print("torch == fp64 path: %.1f %%" % (100 * (t_torch == t_f64).mean()))

r_t = ((im_t - c1).clamp(min=0) / (c2 - c3 * im_t)).numpy()
r_g = np.fromfile("/tmp/pq8.f32", np.float32)
print("r: identical %.1f %%  max rel %.2e" % (100 * (r_g == r_t).mean(), np.nanmax(np.abs(r_g - r_t) / np.maximum(r_t, 1e-30))))
p_t = torch.pow(V, 1 / n).numpy()
p_g = np.fromfile("/tmp/pq9.f32", np.float32)
ok = p_t > 1e-30
print("v^(1/n): identical %.1f %%  rel err mean %.2e max %.2e" % (100 * (p_g == p_t)[ok].mean(), (np.abs(p_g - p_t)[ok] / p_t[ok]).mean(), (np.abs(p_g - p_t)[ok] / p_t[ok]).max()))
