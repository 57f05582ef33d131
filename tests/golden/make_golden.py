"""Generates the golden fixtures under tests/golden/ by importing the reference's own Python
(tools/model.py, tools/export_llama.py) from /root/reference.  Run in the authoring container
only; the reference never travels, only the vectors written here do.

    python tests/golden/make_golden.py

Fixtures (all seeded, all small):
  g1_quant_gemv.npz   quantize_q80 on N(0,0.02) matrices + W8 GEMV answers in fp64 from the
                      formula of cuda/matmul_kernel.cu:70-74, and F.linear on dequantised weights
  g3_tiny_fp32.bin    legacy_export byte layout of a seeded tools/model.py Transformer
  g3_tiny_fp32.npz    its logits for every prefix of a fixed token list (full-sequence forward)
  g4_tiny_q8.bin      legacy_export_quant byte layout of the same model (classifier untied)
  g4_tiny_q8.npz      logits + greedy stream of the model with dequantised weights loaded
  g5_attention.npz    attention-only cases via torch scaled_dot_product_attention in fp64
"""
import io
import os
import sys

import numpy as np
import torch

REF_TOOLS = "/root/reference/tools"
sys.path.insert(0, REF_TOOLS)
from model import ModelArgs, Transformer  # noqa: E402
import export_llama  # noqa: E402
from export_llama import quantize_q80  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
GS = 64


def g1():
    out = {}
    rng = torch.Generator().manual_seed(1234)
    for name, (K, M) in {"a": (64, 128), "b": (96, 64), "c": (128, 256)}.items():
        w = torch.randn(K, M, generator=rng) * 0.02
        x = torch.randn(M, generator=rng)
        q, s, _ = quantize_q80(w, GS)
        q = q.reshape(K, M)
        qd = q.numpy().astype(np.float64)
        sd = s.numpy().astype(np.float64).repeat(GS).reshape(K, M)
        y64 = (x.numpy().astype(np.float64)[None, :] * sd * qd).sum(1)
        deq = (q.float() * s.repeat_interleave(GS).reshape(K, M))
        ylin = torch.nn.functional.linear(x[None, :], deq)[0]
        out[f"{name}_w"] = w.numpy()
        out[f"{name}_x"] = x.numpy()
        out[f"{name}_q"] = q.numpy()
        out[f"{name}_s"] = s.numpy()
        out[f"{name}_y64"] = y64
        out[f"{name}_ylinear"] = ylin.numpy()
    np.savez(os.path.join(HERE, "g1_quant_gemv.npz"), **out)


def tiny_model():
    torch.manual_seed(0)
    args = ModelArgs(dim=128, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=128, hidden_dim=256,
                     max_seq_len=32, norm_eps=1e-5)
    m = Transformer(args)
    m.eval()
    # make the norm weights non-trivial so their placement in the file is exercised
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for layer in m.layers:
            layer.attention_norm.weight.copy_(1.0 + 0.1 * torch.randn(128, generator=g))
            layer.ffn_norm.weight.copy_(1.0 + 0.1 * torch.randn(128, generator=g))
        m.norm.weight.copy_(1.0 + 0.1 * torch.randn(128, generator=g))
        # random-init logits at std 0.02 are nearly flat; scale the residual-stream producers so
        # the greedy argmax has a margin far above any fp32 reordering noise
        m.tok_embeddings.weight.mul_(40.0)
    return m, args


def export_bytes(fn, model, path):
    vs = model.params.vocab_size
    fn(model, path)
    model.params.vocab_size = abs(vs)  # the exporters flip the sign in place
    with open(path, "rb") as f:
        return f.read()


@torch.no_grad()
def g3_g4():
    m, args = tiny_model()
    tokens = torch.randint(0, args.vocab_size, (12,), generator=torch.Generator().manual_seed(42))
    # --- G3 fp32, tied classifier (positive vocab in header) ---
    export_bytes(export_llama.legacy_export, m, os.path.join(HERE, "g3_tiny_fp32.bin"))
    logits = []
    for n in range(1, len(tokens) + 1):
        logits.append(m(tokens[None, :n])[0, -1].numpy())
    # greedy continuation, full-sequence forward every step
    seq = tokens.clone()
    greedy = []
    for _ in range(16):
        nxt = int(torch.argmax(m(seq[None, :])[0, -1]))
        greedy.append(nxt)
        seq = torch.cat([seq, torch.tensor([nxt])])
    np.savez(os.path.join(HERE, "g3_tiny_fp32.npz"), tokens=tokens.numpy().astype(np.int32),
             logits=np.stack(logits), greedy=np.array(greedy, np.int32))

    # --- G4 int8: untie the classifier so the quant exporter emits an int8 classifier ---
    m.output.weight = torch.nn.Parameter(m.tok_embeddings.weight.detach().clone() * 1.0)
    with torch.no_grad():
        m.output.weight.add_(1e-3)  # torch.equal(tok, out) must be False for the exporter
    export_bytes(export_llama.legacy_export_quant, m, os.path.join(HERE, "g4_tiny_q8.bin"))
    # reference-side expected values: the same Transformer with every linear replaced by its
    # dequantised weights (q * scale), embedding left fp32 as in the file
    def deq(p):
        q, s, _ = quantize_q80(p.detach(), GS)
        return (q.float().reshape(-1, GS) * s[:, None]).reshape(p.shape)
    for layer in m.layers:
        a, f = layer.attention, layer.feed_forward
        for lin in (a.wq, a.wk, a.wv, a.wo, f.w1, f.w2, f.w3):
            lin.weight.copy_(deq(lin.weight))
    m.output.weight.copy_(deq(m.output.weight))
    logits = []
    for n in range(1, len(tokens) + 1):
        logits.append(m(tokens[None, :n])[0, -1].numpy())
    seq = tokens.clone()
    greedy = []
    for _ in range(16):
        nxt = int(torch.argmax(m(seq[None, :])[0, -1]))
        greedy.append(nxt)
        seq = torch.cat([seq, torch.tensor([nxt])])
    np.savez(os.path.join(HERE, "g4_tiny_q8.npz"), tokens=tokens.numpy().astype(np.int32),
             logits=np.stack(logits), greedy=np.array(greedy, np.int32))


_CACHES = {}


def sdpa_case(H, KVH, hs, seq_len, pos, Tq, seed, L=2, layer=1):
    g = torch.Generator().manual_seed(seed)
    kv_dim = KVH * hs
    key = (L, seq_len, kv_dim)
    if key not in _CACHES:  # cases of one geometry share one cache: keeps the fixture small
        cg = torch.Generator().manual_seed(9000 + len(_CACHES))
        _CACHES[key] = (len(_CACHES), torch.randn(L, seq_len, kv_dim, generator=cg),
                        torch.randn(L, seq_len, kv_dim, generator=cg))
    cache_id, kc, vc = _CACHES[key]
    q = torch.randn(Tq, H, hs, generator=g)
    n = pos + 1
    k = kc[layer, :n].reshape(n, KVH, hs).repeat_interleave(H // KVH, dim=1).double()
    v = vc[layer, :n].reshape(n, KVH, hs).repeat_interleave(H // KVH, dim=1).double()
    qq = q.double().transpose(0, 1)  # H,Tq,hs
    # query row i is at absolute position pos-(Tq-1)+i and sees cache rows <= that position
    rows = torch.arange(Tq)[:, None] + (pos - (Tq - 1))
    mask = torch.arange(n)[None, :] <= rows
    o = torch.nn.functional.scaled_dot_product_attention(qq, k.transpose(0, 1), v.transpose(0, 1),
                                                         attn_mask=mask)
    return dict(q=q.numpy(), out=o.transpose(0, 1).float().numpy(),
                meta=np.array([H, KVH, hs, seq_len, pos, Tq, L, layer, cache_id], np.int32))


def g5():
    out = {}
    cases = [(8, 2, 64, 128, 0, 1), (8, 2, 64, 128, 1, 1), (8, 2, 64, 128, 31, 1),
             (8, 2, 64, 128, 100, 1), (4, 1, 128, 96, 63, 1), (8, 2, 64, 128, 99, 100),
             (4, 1, 128, 96, 79, 37), (4, 2, 32, 48, 40, 8)]
    for i, (H, KVH, hs, S, pos, Tq) in enumerate(cases):
        for k, v in sdpa_case(H, KVH, hs, S, pos, Tq, seed=100 + i).items():
            out[f"c{i}_{k}"] = v
    for cache_id, kc, vc in _CACHES.values():
        out[f"cache{cache_id}_k"] = kc.numpy()
        out[f"cache{cache_id}_v"] = vc.numpy()
    # the reference test's deterministic inputs (test/test_op/test_flash_attention.cpp:38-81):
    # seq_len=4, head_num=2, head_size=8, kv_mul=1, pos=3, Tq=seq_len, one layer
    n = 4 * 2 * 8
    idx = np.arange(n)
    q = (0.1 * (idx % 10)).astype(np.float32).reshape(4, 2, 8)
    k = (0.1 * ((idx + 1) % 10)).astype(np.float32).reshape(1, 4, 16)
    v = (0.1 * ((idx + 2) % 10)).astype(np.float32).reshape(1, 4, 16)
    kk = torch.tensor(k[0]).reshape(4, 2, 8).double().transpose(0, 1)
    vv = torch.tensor(v[0]).reshape(4, 2, 8).double().transpose(0, 1)
    qq = torch.tensor(q).double().transpose(0, 1)
    o = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, is_causal=True)
    out["ref_q"], out["ref_kc"], out["ref_vc"] = q, k, v
    out["ref_out"] = o.transpose(0, 1).float().numpy()
    out["ref_meta"] = np.array([2, 2, 8, 4, 3, 4, 1, 0, -1], np.int32)
    out["n_cases"] = np.array(len(cases), np.int32)
    np.savez(os.path.join(HERE, "g5_attention.npz"), **out)


if __name__ == "__main__":
    g1()
    g3_g4()
    g5()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
