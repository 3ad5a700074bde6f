"""EDN body of the continuation POST.

The reference publishes with aleph `http/post ... {:content-type :application/edn :form-params msg}`
(reference src/distributed_join/publish.clj:9-11) where msg is
`{:barrier-id barrier-id :messages (map :message joins)}` (core.clj:165-166): a map with two keyword
keys whose second value is a lazy seq, printed by Clojure's printer as a parenthesised list with nil
for missing positions (schema store.clj:75-77).  encode() produces those bytes.
"""

# Clojure's printer for strings with *print-readably*: char-escape-string (core_print.clj)
_ESCAPES = {'"': '\\"', "\\": "\\\\", "\n": "\\n", "\t": "\\t", "\r": "\\r", "\f": "\\f", "\b": "\\b"}


def pr_str(s):
    return '"' + "".join(_ESCAPES.get(ch, ch) for ch in s) + '"'


def encode(barrier_id, messages):
    """messages: list of str or None, in position order."""
    items = " ".join("nil" if m is None else pr_str(m) for m in messages)
    return ("{:barrier-id " + pr_str(barrier_id) + ", :messages (" + items + ")}").encode("utf-8")
