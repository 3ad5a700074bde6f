"""EDN body of the continuation POST.

The reference publishes with aleph `http/post ... {:content-type :application/edn :form-params msg}`
(reference src/distributed_join/publish.clj:9-11) where msg is
`{:barrier-id barrier-id :messages (map :message joins)}` (core.clj:165-166): a map with two keyword
keys whose second value is a lazy seq, printed by Clojure's printer as a parenthesised list with nil
for missing positions (schema store.clj:75-77).  encode() produces those bytes.
"""

CONTENT_TYPE = "application/edn"          # publish.clj:10

# Clojure's printer for strings with *print-readably*: char-escape-string (core_print.clj)
_ESCAPES = {'"': '\\"', "\\": "\\\\", "\n": "\\n", "\t": "\\t", "\r": "\\r", "\f": "\\f", "\b": "\\b"}


def pr_str(s):
    return '"' + "".join(_ESCAPES.get(ch, ch) for ch in s) + '"'


def encode(barrier_id, messages):
    """messages: list of str or None, in position order."""
    items = " ".join("nil" if m is None else pr_str(m) for m in messages)
    return ("{:barrier-id " + pr_str(barrier_id) + ", :messages (" + items + ")}").encode("utf-8")


def decode(body):
    """The inverse of encode() for a receiver (and the tests' destination): bytes -> (barrier-id, [str or None])."""
    text = body.decode("utf-8")
    unescape = {v[1]: k for k, v in _ESCAPES.items()}
    pos = 0

    def expect(lit):
        nonlocal pos
        if not text.startswith(lit, pos):
            raise ValueError(f"expected {lit!r} at {pos}: {text[pos:pos + 20]!r}")
        pos += len(lit)

    def string():
        nonlocal pos
        expect('"')
        out = []
        while text[pos] != '"':
            if text[pos] == "\\":
                out.append(unescape[text[pos + 1]])
                pos += 2
            else:
                out.append(text[pos])
                pos += 1
        pos += 1
        return "".join(out)

    expect("{:barrier-id ")
    barrier_id = string()
    expect(", :messages (")
    messages = []
    while text[pos] != ")":
        if text[pos] == " ":
            pos += 1
        elif text.startswith("nil", pos):
            messages.append(None)
            pos += 3
        else:
            messages.append(string())
    expect(")}")
    if pos != len(text):
        raise ValueError("trailing bytes")
    return barrier_id, messages
