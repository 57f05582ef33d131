"""Randomised model-level sweep (a tool, not a test): random small transformer shapes (head sizes
32 / 64 / 128, grouped kv heads, both RoPE variants, with and without q/k/v biases, fp32 or 16-bit
cache), synthetic weights.  For each: the fused graph-replayed token loop against the
reference-structured loop (op::Layer objects, registry kernels), and batched prompt passes of random
lengths against one fp32 decode step per row.
Every case is printed BEFORE it runs: an out-of-bounds access faults the device and ends the process,
and the last line printed then names the culprit (run it with its output on the terminal, one sweep
per call, AMD_SERIALIZE_KERNEL=3 to have the failing launch named).
Usage: fuzz_model.py [models] [seed]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402,F401
from kx import host  # noqa: E402

def run(N=12, seed=3, head_sizes=(32, 64, 128)):
    """N random models from the generator seeded with `seed`; returns the number of problems."""
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(N):
        hs = int(rng.choice(head_sizes))
        KVH = int(rng.choice([1, 2, 4]))
        H = KVH * int(rng.choice([1, 2, 4]))
        dim = H * hs
        if dim % 64 or (KVH * hs) % 64:   # the loader refuses widths that are not whole quant groups
            continue
        shape = dict(dim=dim, hidden_dim=int(rng.integers(1, 25)) * 64, layer_num=int(rng.integers(1, 4)), head_num=H,
                     kv_head_num=KVH, vocab_size=int(rng.integers(2, 3000)), rope_variant=int(rng.integers(0, 2)),
                     rope_theta=float(rng.choice([1e4, 5e5, 1e6])), rms_eps=float(rng.choice([1e-5, 1e-6])))
        bias = bool(rng.integers(0, 2))
        kv16 = bool(rng.integers(0, 2))
        seq = int(rng.choice([256, 448, 640]))   # the last prompt and its three steps fill the cache to its end
        case = dict(shape, bias=bias, kv16=kv16)
        # the case goes out BEFORE it runs: after a device fault the last line printed names the culprit
        print(f"model {i}: {case}", flush=True)
        try:
            m = host.Model(seq_len=seq, group_size=64, qkv_bias=bias, kv_f16=kv16, **shape)
            ref = host.Model(seq_len=seq, group_size=64, qkv_bias=bias, **shape)
            wseed = int(rng.integers(1, 1 << 30))
            m.init_synthetic(wseed)
            ref.init_synthetic(wseed)
            if bias:
                for layer in range(shape["layer_num"]):
                    b = [(rng.standard_normal(n) * 0.2).astype(np.float32) for n in (dim, KVH * hs, KVH * hs)]
                    m.set_qkv_bias(layer, *b)
                    ref.set_qkv_bias(layer, *b)
            toks = rng.integers(0, shape["vocab_size"], 8).astype(np.int32)
            tol = 4e-3 if kv16 else 2e-4
            for p, t in enumerate(toks):
                a, b = m.step_logits(int(t), p), ref.forward_layers(int(t), p)
                if not np.abs(a - b).max() <= tol * np.abs(b).max():
                    bad += 1
                    print(f"MISMATCH step {p}: {np.abs(a - b).max() / np.abs(b).max():.2e} {case}", flush=True)
                    # the same step again on both models: a result that changes names the path that
                    # is not repeatable (a race, or memory read before it was written)
                    a2, b2 = m.step_logits(int(t), p), ref.forward_layers(int(t), p)
                    sc = np.abs(b2).max()
                    print(f"  again: fused loop moved {np.abs(a2 - a).max() / sc:.2e}, layer loop moved "
                          f"{np.abs(b2 - b).max() / sc:.2e}, now apart {np.abs(a2 - b2).max() / sc:.2e}; "
                          f"worst element {int(np.argmax(np.abs(a - b)))} of {a.size}: {a[np.argmax(np.abs(a - b))]:.5f} vs "
                          f"{b[np.argmax(np.abs(a - b))]:.5f}, elements off {(np.abs(a - b) > tol * np.abs(b).max()).sum()}", flush=True)
                    # where the two caches part: K rows that differ while V rows agree point at the
                    # rotation (tables, position), both at the projections, neither at the attention
                    for layer in range(shape["layer_num"]):
                        (k1, v1), (k2, v2) = m.read_kv(layer, p + 1), ref.read_kv(layer, p + 1)
                        print(f"  layer {layer}: K rows apart {[float(f'{x:.1e}') for x in np.abs(k1 - k2).max(1) / np.abs(k2).max()]}"
                              f" V rows apart {[float(f'{x:.1e}') for x in np.abs(v1 - v2).max(1) / np.abs(v2).max()]}"
                              f" |K| fused {[float(f'{x:.2e}') for x in np.abs(k1).max(1)]} layer loop {[float(f'{x:.2e}') for x in np.abs(k2).max(1)]}", flush=True)
                    break
            for n in (int(rng.integers(2, 65)), int(rng.integers(65, 193)), int(rng.integers(193, seq - 3)), seq - 3):
                prompt = rng.integers(0, shape["vocab_size"], n).astype(np.int32)
                got, want = m.prefill(prompt), ref.prefill(prompt, exact=True)
                if not (np.isfinite(got).all() and np.abs(got - want).max() <= 6e-3 * np.abs(want).max()):
                    bad += 1
                    print(f"MISMATCH prefill {n} rows: {np.abs(got - want).max() / np.abs(want).max():.2e} {case}", flush=True)
                m.decode(3, use_graph=True)
                m.sync()
                print(f"  prompt of {n} rows + 3 steps done", flush=True)
            # the same model over two or four tensor-parallel ranks (objects of this process, reduces
            # through a host rendezvous) where its widths divide
            for tp in (2, 4):
                if H % tp or KVH % tp or (shape["hidden_dim"] // tp) % 64 or shape["hidden_dim"] % tp or \
                        (KVH // tp * hs) % 64 or bias or kv16:
                    continue
                group = host.loopback_create(tp)
                ranks = [host.Model(seq_len=seq, group_size=64, tp_size=tp, tp_rank=r, **shape) for r in range(tp)]
                for r_ in ranks:
                    r_.init_synthetic(wseed)
                    r_.join_loopback(group)
                short = rng.integers(0, shape["vocab_size"], int(rng.integers(2, 193))).astype(np.int32)
                want_tp = ref.prefill(short, exact=True)

                def work(r, out):
                    out[r] = (np.stack([ranks[r].step_logits(int(t), p) for p, t in enumerate(toks[:4])]),
                              ranks[r].prefill(short))
                out = [None] * tp
                ths = [threading.Thread(target=work, args=(r, out)) for r in range(tp)]
                for t_ in ths:
                    t_.start()
                for t_ in ths:
                    t_.join(120)
                want_steps = np.stack([ref.forward_layers(int(t), p) for p, t in enumerate(toks[:4])])
                for r in range(tp):
                    if out[r] is None or not (np.abs(out[r][0] - want_steps).max() <= 2e-4 * np.abs(want_steps).max()
                                              and np.abs(out[r][1] - want_tp).max() <= 6e-3 * np.abs(want_tp).max()):
                        bad += 1
                        print(f"MISMATCH tp={tp} rank {r} {case}", flush=True)
                        break
                for r_ in ranks:
                    r_.close()
                host.loopback_destroy(group)
                print(f"  tp={tp} done", flush=True)
            m.close()
            ref.close()
        except kx.KxError as e:
            bad += 1
            print(f"ERROR {e} {case}", flush=True)
    print(f"{N} models, {bad} problems", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 12,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 3) else 0)
