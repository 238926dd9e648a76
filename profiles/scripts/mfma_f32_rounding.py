"""Which float32 arithmetic does v_mfma_f32_32x32x2_f32 perform?  One bias-free Linear layer of K
input columns through the layer-chain kernel, compared bit by bit with numpy models of
D = C + a0 b0 + a1 b1 per instruction (the products of two float32 are exact in float64):
  fma2      two fused multiply-adds in a row, each rounded to nearest even
  mul_add   products rounded to float32, then added one by one (round to nearest even)
  dot2_rne  C + a0 b0 + a1 b1 summed exactly, rounded once to nearest even
  dot2_rtz  the same sum, rounded once toward zero
usage (GPU box, repo root): python profiles/scripts/mfma_f32_rounding.py"""
import sys

sys.path.insert(0, "multimodal-fusion_amd")
import numpy as np
import torch

from mvfusion import dense as _dense

f32, f64, ld = np.float32, np.float64, np.longdouble


def rtz(x):
    """longdouble -> float32 toward zero."""
    r = x.astype(f32)
    over = np.abs(r.astype(ld)) > np.abs(x)
    return np.where(over, np.nextafter(r, f32(0)), r).astype(f32)


def models(x, w):
    n, K = x.shape
    out = {}
    acc = {k: np.zeros((n, w.shape[0]), f32) for k in ("fma2", "mul_add", "dot2_rne", "dot2_rtz")}
    for k in range(0, K, 2):
        p0 = x[:, k:k + 1].astype(ld) * w[:, k][None, :].astype(ld)
        p1 = x[:, k + 1:k + 2].astype(ld) * w[:, k + 1][None, :].astype(ld)
        a = acc["fma2"]
        a = (a.astype(ld) + p0).astype(f32)
        acc["fma2"] = (a.astype(ld) + p1).astype(f32)
        a = acc["mul_add"]
        a = (a.astype(ld) + p0.astype(f32).astype(ld)).astype(f32)
        acc["mul_add"] = (a.astype(ld) + p1.astype(f32).astype(ld)).astype(f32)
        s = acc["dot2_rne"].astype(ld) + p0 + p1
        acc["dot2_rne"] = s.astype(f32)
        s = acc["dot2_rtz"].astype(ld) + p0 + p1
        acc["dot2_rtz"] = rtz(s)
    return acc


rng = np.random.default_rng(0)
for K in (2, 4, 8, 16):
    x = rng.normal(size=(4096, K)).astype(f32)
    w = rng.normal(size=(8, K)).astype(f32)
    lin = torch.nn.Linear(K, 8, bias=False)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(w))
    lin = lin.to("cuda:0")
    with torch.no_grad():
        layers, xin = _dense.aligned_input([_dense.linear_layer(lin)], torch.from_numpy(x).to("cuda:0"))
        y = _dense.run_layers(layers, xin).cpu().numpy()
    exact = x.astype(f64) @ w.astype(f64).T
    line = [f"K={K:2d}: HIP error rms {np.sqrt(((y - exact) ** 2).mean()):.2e}"]
    for name, m in models(x, w).items():
        line.append(f"{name} bit-equal {float((m.view(np.uint32) == y.view(np.uint32)).mean()):.4f} "
                    f"(rms {np.sqrt(((m - exact) ** 2).mean()):.2e})")
    print(" | ".join(line), flush=True)

# A Linear layer with a bias, in the regime of the pools' score layers (8 inputs of magnitude
# 0.5, weights up to 0.17, a bias that cancels part of the sum): the chain kernel starts its
# accumulators from the bias, the oracle (and a library GEMM) adds it last.
print("with a bias (error rms relative to the rms of the exact result):")
for K, b0 in ((8, 0.1), (8, 0.5), (32, 0.1), (32, 1.0)):
    x = (0.5 * rng.normal(size=(8192, K))).astype(f32)
    w = rng.uniform(-0.17, 0.17, size=(8, K)).astype(f32)
    b = np.full(8, b0, f32)
    lin = torch.nn.Linear(K, 8, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(w))
        lin.bias.copy_(torch.from_numpy(b))
    lin = lin.to("cuda:0")
    with torch.no_grad():
        layers, xin = _dense.aligned_input([_dense.linear_layer(lin)], torch.from_numpy(x).to("cuda:0"))
        y = _dense.run_layers(layers, xin).cpu().numpy()
    exact = x.astype(f64) @ w.astype(f64).T + b.astype(f64)
    first = np.broadcast_to(b, exact.shape).astype(f32)
    last = np.zeros(exact.shape, f32)
    for k in range(K):
        p = x[:, k:k + 1].astype(ld) * w[:, k][None, :].astype(ld)
        first = (first.astype(ld) + p).astype(f32)
        last = (last.astype(ld) + p).astype(f32)
    last = (last + b).astype(f32)
    sc = np.sqrt((exact ** 2).mean())
    r = lambda a: np.sqrt(((a - exact) ** 2).mean()) / sc
    print(f"K={K:2d} bias {b0}: HIP {r(y):.2e} | fma chain from the bias {r(first):.2e} | fma chain, bias last {r(last):.2e}")
