from distributed_join_amd import edn


def test_publish_body_shape():
    # {:barrier-id id :messages (map :message joins)}, core.clj:165-166; nil for a missing position
    assert edn.encode("a~b", ["fffffff", None, "x"]) == b'{:barrier-id "a~b", :messages ("fffffff" nil "x")}'
    assert edn.encode("a~b", []) == b'{:barrier-id "a~b", :messages ()}'
    assert edn.encode("a~b", [""]) == b'{:barrier-id "a~b", :messages ("")}'


def test_string_escapes_follow_the_clojure_printer():
    assert edn.pr_str('say "hi"\\now\n\t') == '"say \\"hi\\"\\\\now\\n\\t"'
    assert edn.encode("j~t", ["é"]) == '{:barrier-id "j~t", :messages ("é")}'.encode("utf-8")


def test_decode_is_the_inverse():
    for bid, msgs in [("a~b", ["fffffff", None, "x"]), ("j~t", []), ("q", [""]), ("w~1", ['say "hi"\\now\n\t', None, "é\u2603", "nil", ")}"])]:
        assert edn.decode(edn.encode(bid, msgs)) == (bid, msgs)
