"""Guided decoding, host side: regex / JSON schema -> byte-level DFA.

The reference asks its vLLM server for constrained output in two ways: ``guided_regex`` (the olmOCR front-matter
pattern, /root/reference/karanta/pipeline.py:304-307) and ``response_format = {"type": "json_schema", ...}``
(/root/reference/karanta/data/utils.py:322-440 through VLLMClient.generate,
/root/reference/bulk_processing/workers/vllm_client.py:155-196).  vLLM compiles both to an automaton over the
tokenizer's vocabulary and masks the logits of every step.  Here:

* this module compiles the pattern to a minimal DFA over BYTES (UTF-8): ``trans[state, byte] -> state`` with state 0
  the dead state, plus the accepting states;
* the device owns the rest (kr_guide.hip): one launch turns the DFA and the vocabulary's byte strings into one
  allowed-token bit mask per DFA state; the sampling pass of every decode step masks the logits of a guided slot with
  the row of its current state, and a tiny kernel walks the sampled token's bytes to the next state — all inside
  the replayed hipGraph, no host in the loop.

A token is allowed in state s iff walking its bytes from s never hits the dead state (every live state can still
reach acceptance: dead-end states are folded into state 0 here); EOS is allowed iff s is accepting.

Supported regex syntax: literals, ``\\`` escapes (``\\n \\t \\r \\f \\v \\d \\D \\w \\W \\s \\S \\xHH \\uHHHH`` and
escaped punctuation), ``.`` (anything but a newline), classes ``[...]`` / ``[^...]`` with ranges, groups ``(...)``,
``(?:...)``, ``(?P<name>...)``, alternation, ``* + ? {m} {m,} {m,n}`` (a trailing ``?`` for laziness is accepted and
meaningless for a full match), ``^`` / ``$`` as no-ops.  Classes are ASCII-based (``\\w`` = ``[A-Za-z0-9_]``); a negated
class, ``.``, ``\\D \\W \\S`` also match every well-formed multi-byte UTF-8 character.  Look-around and back-references
raise :class:`GuideError` (no regular language).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

MAX_STATES = 20000          # DFA states after minimisation the device tables are sized for (uint16 ids)
_ASCII = (1 << 128) - 1


class GuideError(ValueError):
    """Pattern outside the supported syntax, or too large — HTTP 400 at the server."""


# ----------------------------------------------------------------------------- regex -> AST
# AST nodes: ("set", mask256:int) | ("cat", [nodes]) | ("alt", [nodes]) | ("rep", node, lo, hi|None)
def _m(chars: str) -> int:
    v = 0
    for c in chars:
        v |= 1 << ord(c)
    return v


def _rng(a: int, b: int) -> int:
    return ((1 << (b + 1)) - 1) & ~((1 << a) - 1)


_DIGIT = _rng(0x30, 0x39)
_WORD = _DIGIT | _rng(0x41, 0x5A) | _rng(0x61, 0x7A) | _m("_")
_SPACE = _m(" \t\n\r\f\v")
_CONT = ("set", _rng(0x80, 0xBF))
# any well-formed multi-byte UTF-8 character (lead byte decides the length)
_ANY_MB = ("alt", [("cat", [("set", _rng(0xC2, 0xDF)), _CONT]),
                   ("cat", [("set", _rng(0xE0, 0xEF)), _CONT, _CONT]),
                   ("cat", [("set", _rng(0xF0, 0xF4)), _CONT, _CONT, _CONT])])
_EMPTY = ("cat", [])


def _class_node(ascii_mask: int, any_mb: bool, chars: Sequence[str]) -> tuple:
    alts = []
    if ascii_mask:
        alts.append(("set", ascii_mask))
    for c in chars:
        alts.append(("cat", [("set", 1 << b) for b in c.encode("utf-8")]))
    if any_mb:
        alts.append(_ANY_MB)
    if not alts:
        raise GuideError("empty character class")
    return alts[0] if len(alts) == 1 else ("alt", alts)


class _Parser:
    def __init__(self, pattern: str):
        self.s, self.i = pattern, 0

    def error(self, msg: str):
        raise GuideError(f"regex: {msg} at position {self.i} of {self.s!r}")

    def peek(self) -> str:
        return self.s[self.i] if self.i < len(self.s) else ""

    def take(self) -> str:
        c = self.peek()
        if not c:
            self.error("unexpected end")
        self.i += 1
        return c

    def parse(self) -> tuple:
        node = self.alt()
        if self.i != len(self.s):
            self.error("unbalanced ')'")
        return node

    def alt(self) -> tuple:
        branches = [self.cat()]
        while self.peek() == "|":
            self.i += 1
            branches.append(self.cat())
        return branches[0] if len(branches) == 1 else ("alt", branches)

    def cat(self) -> tuple:
        items = []
        while self.peek() not in ("", "|", ")"):
            items.append(self.rep())
        return items[0] if len(items) == 1 else ("cat", items)

    def rep(self) -> tuple:
        node = self.atom()
        while True:
            c = self.peek()
            if c == "*":
                lo, hi = 0, None
            elif c == "+":
                lo, hi = 1, None
            elif c == "?":
                lo, hi = 0, 1
            elif c == "{":
                j = self.s.find("}", self.i)
                body = self.s[self.i + 1:j] if j > 0 else ""
                parts = body.split(",")
                if (j < 0 or not (1 <= len(parts) <= 2) or not all(p.isdigit() or (p == "" and len(parts) == 2) for p in parts)
                        or parts == ["", ""]):
                    break   # a literal '{' (as Python's re treats it)
                lo = int(parts[0] or 0)
                hi = lo if len(parts) == 1 else (int(parts[1]) if parts[1] else None)
                if hi is not None and hi < lo:
                    self.error("bad repeat range")
                self.i = j
            else:
                break
            self.i += 1
            if self.peek() == "?":   # lazy suffix: same language under a full match
                self.i += 1
            node = ("rep", node, lo, hi)
        return node

    def atom(self) -> tuple:
        c = self.take()
        if c == "(":
            if self.peek() == "?":
                self.i += 1
                k = self.take()
                if k == "P" and self.peek() == "<":
                    j = self.s.find(">", self.i)
                    if j < 0:
                        self.error("unterminated group name")
                    self.i = j + 1
                elif k != ":":
                    self.error("look-around / inline flags are not supported")
            node = self.alt()
            if self.take() != ")":
                self.error("expected ')'")
            return node
        if c == "[":
            return self.char_class()
        if c == ".":
            return _class_node(_ASCII & ~_m("\n"), True, ())
        if c in "^$":
            return _EMPTY
        if c == "\\":
            return self.escape(in_class=False)
        if c in "*+?":
            self.error("nothing to repeat")
        return ("cat", [("set", 1 << b) for b in c.encode("utf-8")]) if ord(c) > 127 else ("set", 1 << ord(c))

    _SIMPLE = {"n": "\n", "t": "\t", "r": "\r", "f": "\f", "v": "\v", "a": "\a", "0": "\0"}

    def escape(self, in_class: bool):
        """After a backslash.  Outside a class: a node; inside: (ascii_mask, any_mb, chars)."""
        c = self.take()
        triple = None
        if c == "d":
            triple = (_DIGIT, False, [])
        elif c == "D":
            triple = (_ASCII & ~_DIGIT, True, [])
        elif c == "w":
            triple = (_WORD, False, [])
        elif c == "W":
            triple = (_ASCII & ~_WORD, True, [])
        elif c == "s":
            triple = (_SPACE, False, [])
        elif c == "S":
            triple = (_ASCII & ~_SPACE, True, [])
        elif c in ("x", "u"):
            n = 2 if c == "x" else 4
            h = self.s[self.i:self.i + n]
            if len(h) != n or any(ch not in "0123456789abcdefABCDEF" for ch in h):
                self.error("bad \\x / \\u escape")
            self.i += n
            ch = chr(int(h, 16))
            triple = (1 << ord(ch), False, []) if ord(ch) < 128 else (0, False, [ch])
        elif c in self._SIMPLE:
            triple = (1 << ord(self._SIMPLE[c]), False, [])
        elif c.isalnum():
            self.error(f"unsupported escape \\{c}")   # \b, \B, \1 ... : not regular / not supported
        else:
            triple = (1 << ord(c), False, []) if ord(c) < 128 else (0, False, [c])
        return triple if in_class else _class_node(*triple)

    def char_class(self) -> tuple:
        neg = self.peek() == "^"
        if neg:
            self.i += 1
        mask, any_mb, chars = 0, False, []
        first = True
        while True:
            c = self.take()
            if c == "]" and not first:
                break
            first = False
            if c == "\\":
                m, a, ch = self.escape(in_class=True)
                single = m if (m and m & (m - 1) == 0 and not a and not ch) else None
            else:
                m, a, ch = ((1 << ord(c), False, []) if ord(c) < 128 else (0, False, [c]))
                single = m or None
            # range a-b (both ends single ASCII characters)
            if self.peek() == "-" and self.i + 1 < len(self.s) and self.s[self.i + 1] != "]":
                if single is None:
                    self.error("class range needs ASCII end points")
                self.i += 1
                e = self.take()
                if e == "\\":
                    m2, a2, ch2 = self.escape(in_class=True)
                    if not (m2 and m2 & (m2 - 1) == 0 and not a2 and not ch2):
                        self.error("class range needs ASCII end points")
                    hi = m2.bit_length() - 1
                else:
                    if ord(e) > 127:
                        self.error("class range needs ASCII end points")
                    hi = ord(e)
                lo = single.bit_length() - 1
                if hi < lo:
                    self.error("bad class range")
                mask |= _rng(lo, hi)
                continue
            mask |= m
            any_mb |= a
            chars += ch
        if neg:
            if chars:
                self.error("a negated class may only list ASCII characters")
            return _class_node(_ASCII & ~mask, not any_mb, ())
        return _class_node(mask, any_mb, chars)


# ----------------------------------------------------------------------------- AST -> NFA -> DFA
class _Nfa:
    def __init__(self):
        self.eps: List[List[int]] = []
        self.edge: List[List[Tuple[int, int]]] = []   # (byte mask, target)

    def new(self) -> int:
        self.eps.append([])
        self.edge.append([])
        if len(self.eps) > 100000:
            raise GuideError("pattern too large (repeat counts expand to more than 100k NFA states)")
        return len(self.eps) - 1

    def build(self, node: tuple, a: int) -> int:
        """Adds `node` starting at state a; returns its end state."""
        kind = node[0]
        if kind == "set":
            b = self.new()
            self.edge[a].append((node[1], b))
            return b
        if kind == "cat":
            for n in node[1]:
                a = self.build(n, a)
            return a
        if kind == "alt":
            end = self.new()
            for n in node[1]:
                s = self.new()
                self.eps[a].append(s)
                self.eps[self.build(n, s)].append(end)
            return end
        if kind == "rep":
            _, sub, lo, hi = node
            for _ in range(lo):
                a = self.build(sub, a)
            if hi is None:            # sub*
                s = self.new()
                self.eps[a].append(s)
                e = self.build(sub, s)
                self.eps[e].append(s)
                return s
            end = self.new()
            self.eps[a].append(end)
            for _ in range(hi - lo):  # (sub(sub(...)?)?)?
                a = self.build(sub, a)
                self.eps[a].append(end)
            return end
        raise AssertionError(kind)


@dataclass
class Guide:
    """Minimal byte DFA.  State 0 is dead; every other state can reach an accepting state."""
    trans: np.ndarray      # [S, 256] uint16
    accept: np.ndarray     # [S] bool
    start: int
    pattern: str = ""

    @property
    def n_states(self) -> int:
        return int(self.trans.shape[0])

    def walk(self, state: int, data: bytes) -> int:
        for b in data:
            state = int(self.trans[state, b])
            if state == 0:
                break
        return state

    def fullmatch(self, data: bytes) -> bool:
        return bool(self.accept[self.walk(self.start, data)])

    def viable(self, data: bytes) -> bool:
        """Can `data` still be extended to a match?"""
        return self.walk(self.start, data) != 0


def _bits(mask: int) -> np.ndarray:
    """256-bit integer -> bool[256]."""
    return np.unpackbits(np.frombuffer(mask.to_bytes(32, "little"), np.uint8), bitorder="little").astype(bool)


def _determinise(nfa: _Nfa, start: int, final: int) -> Tuple[np.ndarray, np.ndarray]:
    def closure(states) -> frozenset:
        seen, stack = set(states), list(states)
        while stack:
            for t in nfa.eps[stack.pop()]:
                if t not in seen:
                    seen.add(t)
                    stack.append(t)
        return frozenset(seen)

    s0 = closure([start])
    ids: Dict[frozenset, int] = {s0: 1}
    order = [frozenset(), s0]                      # id 0 = the empty set = dead
    rows: List[np.ndarray] = [np.zeros(256, np.int64)]
    clos_cache: Dict[frozenset, int] = {}
    work = 0
    k = 1
    while k < len(order):
        cur = order[k]
        edges = [e for s in cur for e in nfa.edge[s]]
        row = np.zeros(256, np.int64)
        # byte classes: bytes that lie in the same set of edge masks share a target
        classes = [(1 << 256) - 1]
        for m in set(m for m, _ in edges):
            nxt = []
            for c in classes:
                a, b = c & m, c & ~m
                if a:
                    nxt.append(a)
                if b:
                    nxt.append(b)
            classes = nxt
        work += len(edges) * (len(classes) + 1)
        if work > 40_000_000:
            raise GuideError("pattern too large (determinisation work budget exceeded)")
        for c in classes:
            raw = frozenset(t for m, t in edges if m & c)
            if not raw:
                continue
            tid = clos_cache.get(raw)
            if tid is None:
                tgt = closure(raw)
                tid = ids.get(tgt)
                if tid is None:
                    tid = len(order)
                    ids[tgt] = tid
                    order.append(tgt)
                    if tid > 2 * MAX_STATES:
                        raise GuideError("pattern too large (DFA exceeds the state budget)")
                clos_cache[raw] = tid
            row[_bits(c)] = tid
        rows.append(row)
        k += 1
    trans = np.stack(rows)
    accept = np.array([final in s for s in order], bool)
    return trans, accept


def _prune_and_minimise(trans: np.ndarray, accept: np.ndarray, start: int) -> Tuple[np.ndarray, np.ndarray, int]:
    S = trans.shape[0]
    # live = can reach an accepting state (reverse reachability)
    live = accept.copy()
    while True:
        nxt = live | live[trans].any(axis=1)
        if (nxt == live).all():
            break
        live = nxt
    live[0] = False
    trans = np.where(live[trans], trans, 0)
    trans[~live] = 0
    if not live[start]:
        raise GuideError("pattern matches nothing")
    # Moore partition refinement; block 0 = {dead and non-live states}
    block = np.where(live, np.where(accept, 2, 1), 0).astype(np.int64)
    while True:
        sig = np.concatenate([block[:, None], block[trans]], axis=1)
        _, new = np.unique(sig, axis=0, return_inverse=True)
        new = new.reshape(-1)
        # keep the dead block at id 0
        dead_id = new[0]
        new = np.where(new == dead_id, 0, np.where(new < dead_id, new + 1, new))
        if len(np.unique(new)) == len(np.unique(block)):
            block = new
            break
        block = new
    nb = int(block.max()) + 1
    rep = np.zeros(nb, np.int64)
    rep[block[::-1]] = np.arange(S)[::-1]          # first member of each block
    mtrans = block[trans[rep]]
    maccept = accept[rep] & live[rep]
    mtrans[0] = 0
    return mtrans, maccept, int(block[start])


def compile_regex(pattern: str) -> Guide:
    ast = _Parser(pattern).parse()
    nfa = _Nfa()
    start = nfa.new()
    final = nfa.build(ast, start)
    trans, accept = _determinise(nfa, start, final)
    trans, accept, s0 = _prune_and_minimise(trans, accept, 1)
    if trans.shape[0] > MAX_STATES:
        raise GuideError(f"pattern too large ({trans.shape[0]} DFA states > {MAX_STATES})")
    return Guide(np.ascontiguousarray(trans.astype(np.uint16)), np.ascontiguousarray(accept), s0, pattern)


# ----------------------------------------------------------------------------- JSON schema -> regex
_WS = r"[ \n]?"          # optional whitespace between JSON tokens (one space or newline)
_STRING_INNER = r'(?:[^"\\\x00-\x1f]|\\["\\/bfnrt]|\\u[0-9a-fA-F]{4})'
_STRING = '"' + _STRING_INNER + '*"'
_INTEGER = r"-?(?:0|[1-9][0-9]*)"
_NUMBER = _INTEGER + r"(?:\.[0-9]+)?(?:[eE][+-]?[0-9]+)?"
_BOOLEAN = r"(?:true|false)"
_NULL = r"null"
_RE_SPECIAL = set(".^$*+?{}[]\\|()")


def _lit(text: str) -> str:
    return "".join("\\" + c if c in _RE_SPECIAL else c for c in text)


def _json_const(v: Any) -> str:
    return _lit(json.dumps(v, ensure_ascii=False, separators=(",", ":")))


def _any_json(depth: int) -> str:
    scalar = f"(?:{_STRING}|{_NUMBER}|{_BOOLEAN}|{_NULL})"
    if depth <= 0:
        return scalar
    inner = _any_json(depth - 1)
    arr = rf"\[{_WS}(?:{inner}(?:{_WS},{_WS}{inner})*)?{_WS}\]"
    obj = rf"\{{{_WS}(?:{_STRING}{_WS}:{_WS}{inner}(?:{_WS},{_WS}{_STRING}{_WS}:{_WS}{inner})*)?{_WS}\}}"
    return f"(?:{scalar}|{arr}|{obj})"


def schema_to_regex(schema: Dict[str, Any], _defs: Optional[Dict[str, Any]] = None, _depth: int = 0) -> str:
    """JSON schema -> regex of its serialisations.  Objects emit their properties in declaration order; a property
    outside ``required`` may be left out.  Covers what the reference's schemas use (data/utils.py:322-600: object,
    array, string, integer, number, boolean, null, type lists, enum, const, anyOf/oneOf, $ref into $defs)."""
    if _depth > 24:
        raise GuideError("JSON schema nests too deep (recursive $ref?)")
    if _defs is None:
        _defs = dict(schema.get("$defs") or schema.get("definitions") or {})
    sub = lambda s: schema_to_regex(s, _defs, _depth + 1)
    if not isinstance(schema, dict):
        raise GuideError("JSON schema must be an object")
    if "$ref" in schema:
        name = str(schema["$ref"]).split("/")[-1]
        if name not in _defs:
            raise GuideError(f"unresolved $ref {schema['$ref']!r}")
        return sub(_defs[name])
    if "const" in schema:
        return _json_const(schema["const"])
    if "enum" in schema:
        return "(?:" + "|".join(_json_const(v) for v in schema["enum"]) + ")"
    for key in ("anyOf", "oneOf"):
        if key in schema:
            return "(?:" + "|".join(sub(s) for s in schema[key]) + ")"
    if "allOf" in schema:
        if len(schema["allOf"]) != 1:
            raise GuideError("allOf with more than one member is not supported")
        return sub(schema["allOf"][0])
    ty = schema.get("type")
    if isinstance(ty, list):
        return "(?:" + "|".join(sub({**schema, "type": t}) for t in ty) + ")"
    if ty == "string":
        lo, hi = schema.get("minLength"), schema.get("maxLength")
        if "pattern" in schema:
            p = str(schema["pattern"])
            return '"' + (p[1:] if p.startswith("^") else p).removesuffix("$") + '"'
        if lo is None and hi is None:
            return _STRING
        return '"' + _STRING_INNER + "{" + str(int(lo or 0)) + "," + ("" if hi is None else str(int(hi))) + '}"'
    if ty == "integer":
        return _INTEGER
    if ty == "number":
        return _NUMBER
    if ty == "boolean":
        return _BOOLEAN
    if ty == "null":
        return _NULL
    if ty == "array":
        item = sub(schema["items"]) if isinstance(schema.get("items"), dict) else _any_json(2)
        lo = int(schema.get("minItems", 0))
        hi = schema.get("maxItems")
        if hi is not None and int(hi) < max(lo, 1):
            return rf"\[{_WS}\]" if lo == 0 else ""
        more = f"(?:{_WS},{_WS}{item})"
        if lo == 0:
            tail = "*" if hi is None else "{0," + str(int(hi) - 1) + "}"
            return rf"\[{_WS}(?:{item}{more}{tail})?{_WS}\]"
        tail = "{" + str(lo - 1) + "," + ("" if hi is None else str(int(hi) - 1)) + "}"
        return rf"\[{_WS}{item}{more}{tail}{_WS}\]"
    if ty == "object" or "properties" in schema:
        props = schema.get("properties") or {}
        if not props:
            return _any_json(2) if ty != "object" else rf"\{{{_WS}(?:{_STRING}{_WS}:{_WS}{_any_json(1)}(?:{_WS},{_WS}{_STRING}{_WS}:{_WS}{_any_json(1)})*)?{_WS}\}}"
        required = set(schema.get("required") or [])
        names = list(props)
        member = [f'"{_lit(json.dumps(n, ensure_ascii=False)[1:-1])}"{_WS}:{_WS}{sub(props[n])}' for n in names]
        # members in declaration order; optional ones may be absent.  Built right to left as
        # "this member, then optionally-comma the rest" so that commas only appear between present members.
        def tail_from(i: int, need_comma: bool) -> str:
            if i == len(names):
                return ""
            sep = f"{_WS},{_WS}" if need_comma else ""
            here = sep + member[i] + tail_from(i + 1, True)
            if names[i] in required:
                return here
            skip = tail_from(i + 1, need_comma)
            return f"(?:{here}|{skip})" if skip else f"(?:{here})?"
        n_opt = sum(n not in required for n in names)
        if n_opt > 8:
            raise GuideError("more than 8 optional properties in one object are not supported")
        return rf"\{{{_WS}{tail_from(0, False)}{_WS}\}}"
    if ty is None:
        return _any_json(2)
    raise GuideError(f"unsupported JSON schema type {ty!r}")


def regex_for_request(guided_regex: Optional[str], response_format: Optional[Dict[str, Any]]) -> Optional[str]:
    """The constraint an OpenAI-style request body asks for, as one regex (None: unconstrained)."""
    if guided_regex is not None:
        if not isinstance(guided_regex, str) or not guided_regex:
            raise GuideError("guided_regex must be a non-empty string")
        return guided_regex
    if not response_format:
        return None
    if not isinstance(response_format, dict):
        raise GuideError("response_format must be an object")
    kind = response_format.get("type")
    if kind in (None, "text"):
        return None
    if kind == "json_object":
        return rf"\{{{_WS}(?:{_STRING}{_WS}:{_WS}{_any_json(2)}(?:{_WS},{_WS}{_STRING}{_WS}:{_WS}{_any_json(2)})*)?{_WS}\}}"
    if kind == "json_schema":
        js = response_format.get("json_schema") or {}
        schema = js.get("schema", js if "type" in js or "properties" in js else None)
        if not isinstance(schema, dict):
            raise GuideError("response_format.json_schema.schema is missing")
        return schema_to_regex(schema)
    raise GuideError(f"unsupported response_format type {kind!r}")


# ----------------------------------------------------------------------------- vocabulary bytes
def _gpt2_byte_decoder() -> Dict[str, int]:
    """Inverse of the byte -> printable-unicode table of byte-level BPE vocabularies (GPT-2 / Qwen2)."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, n = {}, 0
    for b in range(256):
        if b in keep:
            table[chr(b)] = b
        else:
            table[chr(256 + n)] = b
            n += 1
    return table


def vocab_bytes_from_hf(tk, vocab_size: int) -> List[bytes]:
    """Byte string of every token id of a `tokenizers.Tokenizer` with a byte-level BPE model; special / added
    tokens get b"" (never allowed under a guide)."""
    dec = _gpt2_byte_decoder()
    out = [b""] * vocab_size
    special = set()
    try:
        special = {int(t) for t in tk.get_added_tokens_decoder()}
    except Exception:
        pass
    for tok, i in tk.get_vocab(with_added_tokens=True).items():
        if i >= vocab_size or i in special:
            continue
        try:
            out[i] = bytes(dec[c] for c in tok)
        except KeyError:
            out[i] = b""
    return out


def pack_vocab(token_bytes: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """-> (offsets int32 [V+1], bytes uint8 [total]) for the device."""
    lens = np.fromiter((len(b) for b in token_bytes), np.int64, len(token_bytes))
    off = np.zeros(len(token_bytes) + 1, np.int32)
    off[1:] = np.cumsum(lens)
    flat = np.frombuffer(b"".join(token_bytes), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return off, flat
