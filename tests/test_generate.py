"""The text front-end (kx/generate.py) against the reference demo's loop (demo/main.cpp:5-47),
restated here one token per step, on a stand-in model with a known next-token rule -- no GPU.
Tokenizers: a SentencePiece model trained here on a few sentences (the reference ships none), and
tokenizer.json vocabularies built with the `tokenizers` package carrying the special tokens that
encode.cpp:97-98 / :173-174 look up."""
import io
import json

import numpy as np
import pytest

from kx import generate as gen

CORPUS = ["hello world this is a tiny corpus", "the quick brown fox jumps over the lazy dog",
          "once upon a time there was a little model", "hello hello world"] * 20


@pytest.fixture(scope="module")
def spe():
    import sentencepiece as spm
    w = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(CORPUS), model_writer=w, vocab_size=96,
                                   model_type="bpe", bos_id=1, eos_id=2, unk_id=0, pad_id=-1, minloglevel=2)
    return gen.SpeTokenizer(model_bytes=w.getvalue())


def demo_loop(predict, tokens, total_steps, is_sentence_ending):
    """demo/main.cpp:11-46 with predict(token, pos, is_prompt) -> next (-1 for a prompt row,
    llama3.cpp:801-813)."""
    prompt_len = len(tokens)
    pos, nxt, is_prompt = 0, -1, True
    words = []
    cur = list(tokens)
    while pos < total_steps:
        if pos < prompt_len - 1:
            nxt = predict(cur[pos], pos, True)
        else:
            is_prompt = False
            cur = [nxt]
            nxt = predict(nxt, pos, False)
        if is_sentence_ending(nxt):
            break
        if is_prompt:
            nxt = cur[pos + 1]
        words.append(nxt)
        pos += 1
    return min(pos, total_steps), words


class RuleModel:
    """host.Model's prefill / decode / history with next = rule(token, pos)."""

    class Cfg:
        seq_len = 64

    def __init__(self, vocab, eos, end_at=None):
        self.vocab, self.eos, self.end_at = vocab, eos, end_at
        self.cfg = self.Cfg()
        self.hist = []
        self.decode_calls = []

    def rule(self, token, pos):
        if self.end_at is not None and pos == self.end_at:
            return self.eos
        t = (31 * token + 7 * pos + 3) % self.vocab
        return t if t != self.eos else (t + 1) % self.vocab

    def prefill(self, tokens, pos0=0, exact=False):
        self.hist = [int(t) for t in tokens]
        self.hist.append(self.rule(self.hist[-1], len(self.hist) - 1))

    def decode(self, steps, use_graph=True):
        assert steps > 0
        self.decode_calls.append(steps)
        for _ in range(steps):
            pos = len(self.hist) - 1
            assert pos < self.cfg.seq_len
            self.hist.append(self.rule(self.hist[-1], pos))

    def history(self, n):
        assert n <= len(self.hist)
        return np.asarray(self.hist[:n], np.int32)


def test_spe_tokenizer_mirrors_the_encode_layer(spe):
    ids = spe.encode("hello world")
    assert ids[0] == spe.sp.bos_id() and spe.sp.eos_id() not in ids  # model.cpp:159: BOS, no EOS
    assert spe.decode(ids) == "hello world"
    assert spe.is_sentence_ending(spe.sp.eos_id()) and not spe.is_sentence_ending(ids[1])
    assert spe.vocab_size() == 96


@pytest.mark.parametrize("chunk", [1, 3, 16])
@pytest.mark.parametrize("end_at", [None, 4, 9, 20])
def test_generate_counts_and_words_as_the_demo_loop(spe, chunk, end_at):
    tokens = spe.encode("once upon a time")
    n = len(tokens)
    for total in (0, 1, n - 2, n - 1, n, n + 1, n + 7, 40):
        m = RuleModel(96, spe.sp.eos_id(), end_at)
        want = demo_loop(lambda t, p, pr: -1 if pr else m.rule(t, p), tokens, total, spe.is_sentence_ending)
        got = gen.generate(m, spe, "once upon a time", total, chunk=chunk)
        assert got == want, (total, chunk, end_at)
        assert all(c <= chunk for c in m.decode_calls)


def test_generate_stops_at_the_context_and_prints(spe):
    m = RuleModel(96, spe.sp.eos_id())
    m.cfg.seq_len = 12
    tokens = spe.encode("hello world")
    out = io.StringIO()
    steps, words = gen.generate(m, spe, "hello world", 1000, need_output=True, out=out)
    # positions 0 .. seq_len - 1 are all a cache of seq_len rows holds
    assert steps == 12 and len(words) == 12 and words[:len(tokens) - 1] == tokens[1:]
    assert out.getvalue() == spe.decode(words) + " "
    with pytest.raises(ValueError):
        gen.generate(m, spe, " ".join(["hello"] * 40), 1000)


def _bpe_json(tmp_path, specials):
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=300, special_tokens=specials, show_progress=False))
    path = tmp_path / "tokenizer.json"
    tok.save(str(path))
    return str(path)


def test_bpe_tokenizers_llama3_and_qwen_flags(tmp_path):
    path = _bpe_json(tmp_path, ["<|begin_of_text|>", "<|end_of_text|>", "<|im_start|>", "<|im_end|>"])
    vocab = json.load(open(path))
    assert vocab["model"]["type"] == "BPE"
    l3 = gen.BpeTokenizer(path)  # encode.cpp:97-98, model.cpp:162: BOS, no EOS
    ids = l3.encode("hello world")
    assert ids[0] == l3.bos_id and l3.eos_id not in ids
    assert l3.decode(ids[1:]) == "hello world"
    assert l3.is_sentence_ending(l3.eos_id)
    qw = gen.BpeTokenizer.qwen(path)  # encode.cpp:173-174, model.cpp:166: no BOS, ends at <|im_end|>
    assert qw.encode("hello world") == ids[1:]
    assert qw.is_sentence_ending(qw.tok.token_to_id("<|im_end|>")) and not qw.is_sentence_ending(l3.eos_id)
    with pytest.raises(ValueError):
        gen.BpeTokenizer(_bpe_json(tmp_path, ["<|im_end|>"]))


def test_export_header_fields(golden_bytes):
    # model.cpp:52-122: seven int32, then the group size in a quant file
    dim, hidden, layers, heads, kv_heads, vocab, seq = gen.read_header(golden_bytes("g4_tiny_q8.bin"))
    assert (dim, hidden, layers, heads, kv_heads, abs(vocab)) == (128, 256, 2, 4, 2, 128) and seq > 0


def test_generate_stepwise_for_fp32_models_is_the_demo_loop(spe):
    # a model without the fused token loop (fp32 weights: group_size 0) is stepped one position at a
    # time through forward_layers, exactly the demo's loop
    class Fp32Rule(RuleModel):
        def __init__(self, *a):
            super().__init__(*a)
            self.cfg.group_size = 0
            self.calls = []

        def forward_layers(self, token, pos):
            self.calls.append((int(token), pos))
            logits = np.zeros(self.vocab, np.float32)
            logits[self.rule(int(token), pos)] = 1.0
            return logits

    tokens = spe.encode("once upon a time")
    n = len(tokens)
    for end_at in (None, 3, 9):
        for total in (0, 2, n - 1, n, n + 6, 30):
            m = Fp32Rule(96, spe.sp.eos_id(), end_at)
            m.cfg = type("Cfg", (), dict(seq_len=64, group_size=0))()
            want = demo_loop(lambda t, p, pr: -1 if pr else m.rule(t, p), tokens, total, spe.is_sentence_ending)
            assert gen.generate(m, spe, "once upon a time", total) == want, (end_at, total)
            assert [p for _, p in m.calls] == list(range(len(m.calls)))  # every position once, in order
