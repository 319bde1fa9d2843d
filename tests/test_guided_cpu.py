"""Guided decoding, host side: the regex -> byte DFA compiler is pinned against Python's `re` (full matches and, by
brute force over a small alphabet, prefix viability); JSON-schema patterns against `json`; the oracle's token-mask rule
against the DFA.  The reference's own pattern (karanta/pipeline.py:304-307) and schema (karanta/data/utils.py:322-374)
are the fixtures."""
import itertools
import json
import random
import re

import numpy as np
import pytest

from karanta_ocr_amd import guided as G
from oracle import qwen2vl_oracle as O

# the pattern the reference's pipeline sends with --guided_decoding (karanta/pipeline.py:304-307)
FRONT_MATTER = (r"---\nprimary_language: (?:[a-z]{2}|null)\nis_rotation_valid: (?:True|False|true|false)\n"
                r"rotation_correction: (?:0|90|180|270)\nis_table: (?:True|False|true|false)\n"
                r"is_diagram: (?:True|False|true|false)\n(?:---|---\n[\s\S]+)")

PATTERNS = [
    r"abc", r"a|b|cd", r"a*b+c?", r"(?:ab)*c", r"[a-c]{2,3}", r"[^a]b", r"\d+\.\d{2}", r"x{0,2}y{2}", r"(a|b)*abb",
    r"\s*\w+\s*", r"a.c", r"[\s\S]+", r"(?P<n>ab|a)b?", r"[a\-c]+", r"[]a]+", r"a{2,}", r"(?:a|)(?:b|)", r"\\n|\n", r"a+?b*?",
    r"(?:0|[1-9][0-9]*)", r"[^\x00-\x1f\"]*", r"é+|[aé]b", r"\u00e9x", r"[A-Za-z_][A-Za-z0-9_]*", r"a{,2}",
]


@pytest.mark.parametrize("pat", PATTERNS + [FRONT_MATTER])
def test_dfa_agrees_with_python_re(pat):
    g = G.compile_regex(pat)
    assert g.trans.dtype == np.uint16 and g.trans.shape[1] == 256 and not g.trans[0].any() and not g.accept[0]
    rx = re.compile(pat, re.ASCII)      # the compiler's \\w \\d \\s are the ASCII classes (module docstring)
    rng = random.Random(sum(map(ord, pat)))
    alphabet = "abcdxy019._- \n\"\\é" + "".join(sorted(set(c for c in pat if c.isalnum())))[:12]
    n_match = 0
    for _ in range(3000):
        s = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 7)))
        want = rx.fullmatch(s) is not None
        assert g.fullmatch(s.encode("utf-8")) == want, (pat, s)
        n_match += want
    # strings grown along the DFA are full matches for `re` as well (covers long patterns random strings never hit)
    for _ in range(60):
        st, out = g.start, bytearray()
        for _ in range(200):
            if g.accept[st] and rng.random() < 0.3:
                break
            nxt = np.flatnonzero(g.trans[st])
            if nxt.size == 0:
                break
            ascii_nxt = nxt[nxt < 128]
            b = int(rng.choice(list(ascii_nxt if ascii_nxt.size else nxt)))
            out.append(b)
            st = int(g.trans[st, b])
        if g.accept[st]:
            try:
                text = bytes(out).decode("utf-8")
            except UnicodeDecodeError:
                continue
            assert rx.fullmatch(text) is not None, (pat, text)
            n_match += 1
    assert n_match > 0, f"no positive example exercised for {pat!r}"


@pytest.mark.parametrize("pat", [r"(a|b)*abb", r"a{2}b?|ba", r"[ab]{1,3}c", r"ab*(?:c|ca)"])
def test_prefix_viability_by_brute_force(pat):
    """viable(prefix) <=> some completion matches (every non-dead state can reach acceptance)."""
    g = G.compile_regex(pat)
    rx = re.compile(pat)
    words = ["".join(w) for n in range(0, 9) for w in itertools.product("abc", repeat=n)]
    matches = [w for w in words if rx.fullmatch(w)]
    for w in (w for w in words if len(w) <= 4):     # any viable prefix of <= 4 letters completes within 8
        assert g.viable(w.encode()) == any(m.startswith(w) for m in matches), (pat, w)


def test_minimal_and_bounded():
    assert G.compile_regex(r"(a|b)*abb").n_states == 5           # 4 live states + the dead state
    assert G.compile_regex(FRONT_MATTER).n_states < 200
    for bad in (r"(?=a)b", r"a\1", r"(a", r"a)", r"*a", r"[b-a]", r"\bword", r"[^é]"):
        with pytest.raises(G.GuideError):
            G.compile_regex(bad)
    with pytest.raises(G.GuideError):
        G.compile_regex(r"(?:[a-z]{1,50}){1,50}x{1,400}" * 8)    # blows the state budget instead of the memory


def test_front_matter_document():
    g = G.compile_regex(FRONT_MATTER)
    doc = ("---\nprimary_language: sw\nis_rotation_valid: True\nrotation_correction: 90\nis_table: false\n"
           "is_diagram: False\n---\nHabari ya dunia — ✓ naïve\n\nline two")
    assert g.fullmatch(doc.encode("utf-8")) and re.fullmatch(FRONT_MATTER, doc)
    assert g.viable(doc[:57].encode()) and not g.fullmatch(doc[:57].encode())
    assert not g.viable(b"---\nprimary_language: swa")


REFERENCE_SCHEMA = {   # shape of openai_response_format_schema() (karanta/data/utils.py:322-374), descriptions dropped
    "type": "object",
    "properties": {
        "primary_language": {"type": ["string", "null"]},
        "is_rotation_valid": {"type": "boolean"},
        "rotation_correction": {"type": "integer", "enum": [0, 90, 180, 270], "default": 0},
        "is_table": {"type": "boolean"},
        "is_diagram": {"type": "boolean"},
        "natural_text": {"type": ["string", "null"]},
    },
    "additionalProperties": False,
    "required": ["primary_language", "is_rotation_valid", "rotation_correction", "is_table", "is_diagram", "natural_text"],
}


def test_json_schema_patterns():
    rx = G.regex_for_request(None, {"type": "json_schema", "json_schema": {"name": "page_response", "schema": REFERENCE_SCHEMA,
                                                                           "strict": True}})
    g = G.compile_regex(rx)
    good = {"primary_language": "en", "is_rotation_valid": True, "rotation_correction": 270, "is_table": False,
            "is_diagram": False, "natural_text": "Line \"one\"\n\\ two é \u2713"}
    for dump in (lambda d: json.dumps(d), lambda d: json.dumps(d, ensure_ascii=False), lambda d: json.dumps(d, separators=(",", ":"))):
        text = dump(good)
        assert g.fullmatch(text.encode("utf-8")), text
        assert json.loads(text) == good
    assert g.fullmatch(json.dumps({**good, "natural_text": None}).encode())
    for bad in ({**good, "rotation_correction": 45}, {**good, "is_table": "no"}, {k: v for k, v in good.items() if k != "is_diagram"}):
        assert not g.fullmatch(json.dumps(bad).encode())
    assert not g.fullmatch(b'{"primary_language": "a\nb"')            # raw control characters are not JSON
    # everything the DFA accepts parses as JSON with the schema's keys: walk random accepting paths
    rng = random.Random(5)
    for _ in range(40):
        st, out = g.start, bytearray()
        for _ in range(400):
            if g.accept[st]:
                break
            nxt = np.flatnonzero(g.trans[st][:128])
            b = int(rng.choice(list(nxt)))
            out.append(b)
            st = int(g.trans[st, b])
        if g.accept[st]:
            assert list(json.loads(bytes(out).decode())) == list(REFERENCE_SCHEMA["properties"])

    # arrays, nesting, optional members, $ref, anyOf, const
    schema = {"type": "object", "$defs": {"p": {"type": "object", "properties": {"x": {"type": "number"}}, "required": ["x"]}},
              "properties": {"pts": {"type": "array", "items": {"$ref": "#/$defs/p"}, "minItems": 1, "maxItems": 3},
                             "tag": {"anyOf": [{"const": "a"}, {"type": "null"}]}, "note": {"type": "string", "maxLength": 4}},
              "required": ["pts"]}
    g2 = G.compile_regex(G.schema_to_regex(schema))
    ok = [{"pts": [{"x": 1}]}, {"pts": [{"x": -1.5e3}, {"x": 0}], "tag": "a"}, {"pts": [{"x": 2}], "note": "abcd"},
          {"pts": [{"x": 2}], "tag": None, "note": ""}]
    no = [{"pts": []}, {"pts": [{"x": 1}] * 4}, {"pts": [{"x": 1}], "note": "abcde"}, {"tag": "a"}, {"pts": [{"x": "1"}]}]
    for d in ok:
        assert g2.fullmatch(json.dumps(d).encode()), d
    for d in no:
        assert not g2.fullmatch(json.dumps(d).encode()), d
    assert G.regex_for_request(None, None) is None and G.regex_for_request(None, {"type": "text"}) is None
    assert G.regex_for_request("ab+", {"type": "json_object"}) == "ab+"
    gobj = G.compile_regex(G.regex_for_request(None, {"type": "json_object"}))
    assert gobj.fullmatch(b'{"a": [1, {"b": null}], "c": "d"}') and not gobj.fullmatch(b"[1]")
    with pytest.raises(G.GuideError):
        G.regex_for_request(None, {"type": "json_schema"})
    with pytest.raises(G.GuideError):
        G.regex_for_request(None, {"type": "grammar"})


def test_oracle_token_mask_rule_and_vocab_packing():
    g = G.compile_regex(r"ab*(?:c|ca)")
    vocab = [b"a", b"b", b"c", b"ab", b"bb", b"ca", b"", b"cab", b"d", b""]   # ids 6 and 9: special tokens, 9 = EOS
    eos = [9]
    m0 = O.guide_token_mask(g.trans, g.accept, g.start, vocab, eos)
    assert m0.tolist() == [True, False, False, True, False, False, False, False, False, False]
    s1 = O.guide_walk(g.trans, g.start, b"ab")
    m1 = O.guide_token_mask(g.trans, g.accept, s1, vocab, eos)
    assert m1.tolist() == [False, True, True, False, True, True, False, False, False, False]
    s2 = O.guide_walk(g.trans, s1, b"c")                       # "abc": accepting, may still grow to "abca"
    m2 = O.guide_token_mask(g.trans, g.accept, s2, vocab, eos)
    assert m2.tolist() == [True, False, False, False, False, False, False, False, False, True]
    assert not O.guide_token_mask(g.trans, g.accept, 0, vocab, eos).any()
    off, flat = G.pack_vocab(vocab)
    assert off.dtype == np.int32 and off.tolist() == [0, 1, 2, 3, 5, 7, 9, 9, 12, 13, 13] and bytes(flat) == b"abcabbbcacabd"
    ids, lp = O.top_logprobs(np.array([0.0, 2.0, 2.0, -1.0]), 3)
    assert ids.tolist() == [1, 2, 0] and np.isclose(np.exp(O.log_softmax(np.array([0.0, 2.0, 2.0, -1.0]))).sum(), 1.0)
    assert np.allclose(lp, O.log_softmax(np.array([0.0, 2.0, 2.0, -1.0]))[[1, 2, 0]])


def test_gpt2_byte_table_roundtrip():
    dec = G._gpt2_byte_decoder()
    assert len(dec) == 256 and sorted(dec.values()) == list(range(256))
    assert dec["Ġ"] == 0x20 and dec["Ċ"] == 0x0A and dec["a"] == ord("a")


def test_vocab_bytes_of_a_byte_level_bpe_tokenizer():
    """vocab_bytes_from_hf on a real `tokenizers` byte-level BPE (the kind Qwen2-VL ships): every ordinary token's bytes
    are what decoding that token alone yields, concatenated pieces give back the text, added special tokens are b""."""
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    corpus = ["---\nprimary_language: en\nis_table: False\n", "Habari ya dunia — naïve café ✓ 12345", "{\"a\": true, \"b\": [1, 2]}"] * 20
    tk.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=400, special_tokens=["<|im_start|>", "<|im_end|>"],
                                                      initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    V = tk.get_vocab_size()
    voc = G.vocab_bytes_from_hf(tk, V + 3)                      # a model's vocab may be padded past the tokenizer's
    assert len(voc) == V + 3 and voc[V:] == [b"", b"", b""]
    assert voc[tk.token_to_id("<|im_start|>")] == b"" and voc[tk.token_to_id("<|im_end|>")] == b""
    assert sorted(b for b in voc if len(b) == 1) == [bytes([i]) for i in range(256)], "all 256 single bytes are tokens"
    for text in corpus[:3] + ["naïve ✓\n\n"]:
        ids = tk.encode(text, add_special_tokens=False).ids
        assert b"".join(voc[i] for i in ids) == text.encode("utf-8")
    multi = [i for i in range(V) if len(voc[i]) > 1]
    assert multi, "training produced merges"
    for i in multi[:50]:
        try:
            piece = voc[i].decode("utf-8")
        except UnicodeDecodeError:
            continue                                            # a token may end inside a UTF-8 character
        assert tk.decode([i]) == piece
    # and a guide over that vocabulary: the oracle's rule allows exactly the tokens that keep "---\n" viable
    g = G.compile_regex(r"---\n[a-z_]+: (?:en|null)")
    ok = O.guide_token_mask(g.trans, g.accept, g.start, voc, [])
    assert ok.any() and all(b"---\n".startswith(voc[i]) or voc[i].startswith(b"---\n") for i in np.flatnonzero(ok))


def test_random_patterns_against_python_re():
    """Seeded fuzz: 600 random patterns from a small grammar (classes, escapes, groups, alternation, every quantifier
    form, a non-ASCII literal) — the DFA and `re.fullmatch` (ASCII classes) agree on random strings."""
    rng = random.Random(20251031)
    atoms = ["a", "b", "c", "0", r"\d", r"\w", r"\s", ".", "[ab]", "[^a]", "[a-c0]", r"\.", "é", r"[\s\S]", r"\n", "-", "[-a]", r"[a\]]"]

    def gen(depth=0):
        r = rng.random()
        if depth > 3 or r < 0.35:
            return rng.choice(atoms)
        if r < 0.55:
            return gen(depth + 1) + gen(depth + 1)
        if r < 0.7:
            return "(?:" + gen(depth + 1) + "|" + gen(depth + 1) + ")"
        if r < 0.9:
            return "(?:" + gen(depth + 1) + ")" + rng.choice(["*", "+", "?", "{2}", "{1,3}", "{0,2}", "{2,}", "*?", "+?"])
        return "(" + gen(depth + 1) + ")"

    checked = 0
    while checked < 600:
        pat = gen()
        rx = re.compile(pat, re.ASCII)
        g = G.compile_regex(pat)
        checked += 1
        for _ in range(40):
            s = "".join(rng.choice("abc0 .-\né]x") for _ in range(rng.randint(0, 6)))
            assert g.fullmatch(s.encode()) == (rx.fullmatch(s) is not None), (pat, s)
