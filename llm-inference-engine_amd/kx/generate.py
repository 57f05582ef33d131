"""Text front-end of the token loop: the reference's demo (demo/main.cpp:5-75) over host.Model.

    python -m kx.generate CHECKPOINT TOKENIZER [--family llama2|llama3|qwen2] [--prompt hello]
                          [--steps 128] [--seq-len N]

generate() keeps the demo's accounting -- `total_steps` counts positions, the prompt's rows from the
second one on and every sampled token are the words, the sentence-ending token stops the loop and
is not a word (demo/main.cpp:18-41) -- but not its shape: the demo asks the host for one token per
step; here the prompt is one prefill() and the sampled tokens come from graph-replayed decode()
calls of `chunk` steps whose tokens stay on the device until the chunk is read back and searched
for the sentence end (tokens past it are dropped).

Tokenizers (kuiper/source/op/encode.cpp): SentencePiece for the llama2 family (:24-51, BOS in
front, model.cpp:158-159), the tokenizer.json byte-pair vocabularies of Llama-3 (:62-143, BOS
<|begin_of_text|>, ends at <|end_of_text|>) and Qwen2 (:146-185, no BOS, ends at <|im_end|>)
through the `tokenizers` package instead of the reference's tiktoken port.
"""
import argparse
import struct
import sys
import time

import numpy as np

# rope_variant / rope_theta / rms_eps per family: compile-time switches in the reference
# (LLAMA3_SUPPORT / QWEN2_SUPPORT, kuiper/source/op/kernels/cuda/rope_kernel.cu)
FAMILIES = {
    "llama2": dict(rope_variant=0, rope_theta=1e4, rms_eps=1e-5, tokenizer="spe"),
    "llama3": dict(rope_variant=1, rope_theta=5e5, rms_eps=1e-5, tokenizer="bpe"),
    "qwen2": dict(rope_variant=1, rope_theta=1e6, rms_eps=1e-6, tokenizer="qwen"),
}


class SpeTokenizer:
    """SpeEncodeLayer (encode.cpp:24-59): SentencePiece ids, BOS in front, no EOS behind."""

    def __init__(self, model_path=None, model_bytes=None, has_bos=True, has_eos=False):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=model_path) if model_bytes is None \
            else spm.SentencePieceProcessor(model_proto=model_bytes)
        self.has_bos, self.has_eos = has_bos, has_eos

    def encode(self, sentence):
        ids = list(self.sp.encode(sentence))
        if self.has_bos:
            ids.insert(0, self.sp.bos_id())
        if self.has_eos:
            ids.append(self.sp.eos_id())
        return ids

    def decode(self, ids):
        return self.sp.decode([int(i) for i in ids])

    def is_sentence_ending(self, token_id):
        return int(token_id) == self.sp.eos_id()

    def vocab_size(self):
        return self.sp.get_piece_size()


class BpeTokenizer:
    """BpeEncodeLayer / QwenEncodeLayer (encode.cpp:62-185) over a tokenizer.json."""

    def __init__(self, json_path, bos="<|begin_of_text|>", eos="<|end_of_text|>", has_bos=True, has_eos=False):
        from tokenizers import Tokenizer
        self.tok = Tokenizer.from_file(json_path)
        self.bos_id = self.tok.token_to_id(bos)
        self.eos_id = self.tok.token_to_id(eos)
        if self.eos_id is None or (has_bos and self.bos_id is None):
            raise ValueError(f"{json_path} has no {bos!r} / {eos!r}")
        self.has_bos, self.has_eos = has_bos, has_eos

    @classmethod
    def qwen(cls, json_path):
        return cls(json_path, bos="<|im_start|>", eos="<|im_end|>", has_bos=False)

    def encode(self, sentence):
        ids = list(self.tok.encode(sentence, add_special_tokens=False).ids)
        if self.has_bos:
            ids.insert(0, self.bos_id)
        if self.has_eos:
            ids.append(self.eos_id)
        return ids

    def decode(self, ids):
        return self.tok.decode([int(i) for i in ids], skip_special_tokens=False)

    def is_sentence_ending(self, token_id):
        return int(token_id) == self.eos_id

    def vocab_size(self):
        return self.tok.get_vocab_size()


def generate(model, tokenizer, sentence, total_steps, need_output=False, chunk=16, out=sys.stdout):
    """demo/main.cpp:5-47.  Returns (steps, words): the positions done and the token ids behind them
    (prompt rows from the second on, then the sampled tokens up to, not including, a sentence end)."""
    tokens = tokenizer.encode(sentence)
    if not tokens:
        raise ValueError("The tokens is empty.")
    n = len(tokens)
    words = []
    if getattr(getattr(model, "cfg", None), "group_size", 1) == 0:
        # fp32 weights (the demo's own default, a legacy export): no fused token loop -- the
        # reference-structured step through op::Layer and the kernel registry, one position at a time
        pos, words = _generate_stepwise(model, tokenizer, tokens, total_steps)
    elif total_steps <= n - 1:  # the loop would end inside the prompt
        words = tokens[1:total_steps + 1]
        pos = max(total_steps, 0)
    else:
        want = total_steps - (n - 1)  # sampled tokens the loop has room for
        seq_len = getattr(getattr(model, "cfg", None), "seq_len", None)
        if seq_len is not None:  # position n - 1 + k is the last one a cache of seq_len rows holds
            if n > seq_len:
                raise ValueError(f"prompt of {n} tokens in a context of {seq_len}")
            want = min(want, seq_len - n + 1)
        model.prefill(np.asarray(tokens, np.int32))  # leaves the first sampled token behind the prompt
        words = tokens[1:]
        pos = n - 1
        have = 1  # sampled tokens on the device
        seen = 0  # ... of which looked at
        ended = False
        while not ended and seen < want:
            if seen == have:
                step = min(chunk, want - have)
                model.decode(step, use_graph=True)
                have += step
            fresh = model.history(n + have)[n + seen:]
            for t in fresh.tolist():
                if tokenizer.is_sentence_ending(t):
                    ended = True
                    break
                words.append(t)
                pos += 1
                seen += 1
    if need_output:
        out.write(tokenizer.decode(words) + " ")
        out.flush()
    return min(pos, total_steps), words


def _generate_stepwise(model, tokenizer, tokens, total_steps):
    """demo/main.cpp:18-41 as written: forward at every position, the sampler's pick from the last
    prompt position on (argmax_sampler.cpp:6-8 on the host here)."""
    n, pos, nxt, words = len(tokens), 0, -1, []
    total_steps = min(total_steps, model.cfg.seq_len)
    while pos < total_steps:
        if pos < n - 1:
            model.forward_layers(tokens[pos], pos)
            nxt = tokens[pos + 1]
        else:
            logits = model.forward_layers(tokens[pos] if pos == n - 1 else nxt, pos)
            nxt = int(np.argmax(logits))
            if tokenizer.is_sentence_ending(nxt):
                break
        words.append(nxt)
        pos += 1
    return pos, words


def read_header(blob):
    """dim, hidden_dim, layers, heads, kv heads, vocabulary (negative: classifier of its own), seq_len
    (model.cpp:52-122); quant files carry the group size behind them (tools/export_llama.py:134-150)."""
    return struct.unpack("<7i", blob[:28])


def open_model(checkpoint, family, quant=True, seq_len=None, **kw):
    from . import host
    blob = open(checkpoint, "rb").read()
    dim, hidden, layers, heads, kv_heads, vocab, file_seq = read_header(blob)
    gs = struct.unpack("<i", blob[28:32])[0] if quant else 0
    fam = FAMILIES[family]
    m = host.Model(seq_len=seq_len or file_seq, group_size=gs, dim=dim, hidden_dim=hidden, layer_num=layers,
                   head_num=heads, kv_head_num=kv_heads, vocab_size=abs(vocab), rope_variant=fam["rope_variant"],
                   rope_theta=fam["rope_theta"], rms_eps=fam["rms_eps"], **kw)
    m.load_blob(blob)
    return m


def open_tokenizer(path, family):
    kind = FAMILIES[family]["tokenizer"]
    if kind == "spe":
        return SpeTokenizer(path)
    return BpeTokenizer.qwen(path) if kind == "qwen" else BpeTokenizer(path)


def main(argv=None):
    ap = argparse.ArgumentParser(description="checkpoint + tokenizer -> text (demo/main.cpp:50-75)")
    ap.add_argument("checkpoint")
    ap.add_argument("tokenizer")
    ap.add_argument("--family", choices=sorted(FAMILIES), default="llama2")
    ap.add_argument("--fp32", action="store_true",
                    help="legacy fp32 export (one reference-structured step per position) instead of the int8 one")
    ap.add_argument("--prompt", default="hello")
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--seq-len", type=int, default=None)
    a = ap.parse_args(argv)
    model = open_model(a.checkpoint, a.family, quant=not a.fp32, seq_len=a.seq_len)
    tok = open_tokenizer(a.tokenizer, a.family)
    print("Generating...", flush=True)
    t0 = time.perf_counter()
    steps, _ = generate(model, tok, a.prompt, a.steps, need_output=True)
    model.sync()
    print(f"\nsteps/s:{steps / (time.perf_counter() - t0):f}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
