"""How often is the device's fp64 quotient not the nearest double?  Random operands (generic, and with divisors within a
few ulps of one), torch on the GPU against numpy on the host, then the one pair known to be rounded the wrong way.
Random operands never show it; see xdiv() in kernels.hip for why the engine repairs every quotient anyway."""
import json, sys, time
import numpy as np, torch
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
total = bad = 0
worst = None
t0 = time.time()
for rep in range(20):
    a = rng.uniform(-2.0, 2.0, 10_000_000); b = rng.uniform(0.5, 2.0, 10_000_000) * rng.choice([-1.0, 1.0], 10_000_000)
    if rep % 2: b = 1.0 - rng.uniform(0, 1e-12, 10_000_000)   # divisors next to 1, as in a tableau after many pivots
    q_cpu = a / b
    q_gpu = (torch.from_numpy(a).cuda() / torch.from_numpy(b).cuda()).cpu().numpy()
    d = np.nonzero(q_cpu != q_gpu)[0]
    total += a.size; bad += d.size
    if d.size and worst is None: worst = (float(a[d[0]]).hex(), float(b[d[0]]).hex(), float(q_cpu[d[0]]).hex(), float(q_gpu[d[0]]).hex())
# divisors within a few ulps of +-1 (pivot elements of a long run look like this)
near = near_bad = 0
for k in range(1, 33):
    for b0 in (1.0 - k * 2.0 ** -53, 1.0 + k * 2.0 ** -52, -(1.0 - k * 2.0 ** -53), -(1.0 + k * 2.0 ** -52)):
        a = rng.uniform(-2.0, 2.0, 1_000_000); b = np.full_like(a, b0)
        q_cpu = a / b
        q_gpu = (torch.from_numpy(a).cuda() / torch.from_numpy(b).cuda()).cpu().numpy()
        d = np.nonzero(q_cpu != q_gpu)[0]
        near += a.size; near_bad += d.size
        if d.size and worst is None: worst = (float(a[d[0]]).hex(), float(b0).hex(), float(q_cpu[d[0]]).hex(), float(q_gpu[d[0]]).hex())
print(json.dumps({"near_one_quotients": near, "near_one_not_nearest": int(near_bad), "near_one_rate": near_bad / near}))
a0, b0 = float.fromhex("-0x1.6666666666663p-1"), float.fromhex("-0x1.ffffffffffffbp-1")
qg = float((torch.tensor([a0], dtype=torch.float64).cuda() / torch.tensor([b0], dtype=torch.float64).cuda()).cpu()[0])
print(json.dumps({"known_pair": [a0.hex(), b0.hex()], "host": (a0 / b0).hex(), "device": qg.hex()}))
print(json.dumps({"quotients": total, "not_nearest": int(bad), "rate": bad / total, "example": worst, "secs": time.time() - t0}))
