"""The C ABI library loads without a GPU and exports every function include/dj_engine.h declares."""
import os
import re

from distributed_join_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="dj_engine.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dj_[a-z_0-9]+)\s*\(", text)) - {"dj_status_cb"})


def test_header_functions_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 19
    lib = capi.load()
    for n in names:
        assert getattr(lib, n) is not None, n
    assert sorted(capi.SYMBOLS) == names


def test_batcher_header_functions_are_exported_and_bound():
    names = declared_functions("dj_batcher.h")
    lib = capi.load()
    for n in names:
        assert getattr(lib, n) is not None, n
    assert sorted(capi.BATCHER_SYMBOLS) == names


def test_record_layouts_match_header_sizes():
    # sizes stated in include/dj_engine.h comments: 32-byte fork/join records, 64-byte continuation
    assert capi.FORK_REC.itemsize == 32
    assert capi.JOIN_REC.itemsize == 32
    assert capi.CONT.itemsize == 64
    assert capi.CONT.fields["pos_base"][1] == 48


def test_fingerprint_is_pure_and_avoids_reserved_values():
    a = capi.fingerprint("a~b")
    assert a == capi.fingerprint(b"a~b")
    assert a != capi.fingerprint("a~c")
    many = capi.fingerprint_many([f"j{i:x}~t{i:x}" for i in range(20000)])
    assert len({(int(x), int(y)) for x, y in many}) == 20000
    assert int(many[:, 0].min()) >= 2 and int(many[:, 1].min()) >= 1
    assert tuple(int(v) for v in many[7]) == capi.fingerprint("j7~t7")
    # ids that differ only in where the separator sits, and empty / long ids
    assert capi.fingerprint("ab~c") != capi.fingerprint("a~bc")
    assert capi.fingerprint("") != capi.fingerprint("\0")
    assert capi.fingerprint("x" * 511) != capi.fingerprint("x" * 510)


def test_owner_is_stable_and_in_range():
    fps = capi.fingerprint_many([f"j{i}~t{i}" for i in range(4096)])
    for n in (1, 2, 4, 8):
        owners = [capi.owner_of(lo, hi, n) for lo, hi in fps[:512]]
        assert all(0 <= o < n for o in owners)
        if n > 1:
            assert len(set(owners)) == n
