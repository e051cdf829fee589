"""INTEGRATION.md is part of the boundary: a maintainer of the reference pastes its ctypes stub.  CPU: every
`ctypes.Structure` the document declares carries the member list (names, order, C types) of the struct of the same name in
include/lsf_hip.h, and the ABI version / hash the stub checks are the header's.  GPU: the fenced stub of section 3 is
extracted, executed as it stands and held to the reference's five `warp_field_advanced` known answers
(/root/reference tests/test_field_warping.py:25-250, stored as data in tests/golden/ref_test_literals.npz) -- the call it
replaces is slavcheva_optimizer2d.py:224-234."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HEADER = os.path.join(ROOT, "include", "lsf_hip.h")

C_TYPES = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float,
           "double": ctypes.c_double, "uint8_t": ctypes.c_uint8}


def python_blocks():
    return re.findall(r"```python\n(.*?)```", open(DOC).read(), flags=re.S)


def stub_source():
    blocks = [b for b in python_blocks() if "ctypes.CDLL" in b]
    assert len(blocks) == 1, "INTEGRATION.md section 3 holds exactly one binding stub"
    return blocks[0]


def header_struct_members(name):
    """[(member, C type, array length or None)] of `typedef struct <name> { ... } <name>;`, comments stripped"""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S).group(1)
    members = []
    for statement in body.split(";"):
        statement = " ".join(statement.split())
        if not statement:
            continue
        m = re.match(r"(?:const\s+)?([A-Za-z_0-9]+(?:\s*\*)?)\s*(.*)$", statement)
        ctype, names = m.group(1).replace(" ", ""), m.group(2)
        for n in names.split(","):
            n = n.strip()
            arr = re.match(r"(\w+)\[(\w+)\]$", n)
            members.append((arr.group(1), ctype, arr.group(2)) if arr else (n.lstrip("*"), ctype if "*" not in n else
                                                                            ctype + "*", None))
    return members


def test_every_structure_in_the_document_has_the_header_s_member_list():
    source = stub_source()
    # evaluate only the class statements of the stub (no library is loaded, nothing is called)
    classes = re.findall(r"^class (\w+)\(ctypes\.Structure\):.*?(?=^\S)", source, flags=re.S | re.M)
    assert "lsf_grid" in classes
    for name in classes:
        text = re.search(r"^class %s\(ctypes\.Structure\):.*?(?=^\S)" % name, source, flags=re.S | re.M).group(0)
        scope = {"ctypes": ctypes}
        exec(text, scope)
        declared = scope[name]._fields_
        expected = header_struct_members(name)
        assert [f[0] for f in declared] == [m[0] for m in expected], name
        for (field, ftype), (member, ctype, arr) in zip(declared, expected):
            assert arr is None and ftype is C_TYPES[ctype], (name, field)
        assert ctypes.sizeof(scope[name]) == sum(ctypes.sizeof(C_TYPES[m[1]]) for m in expected)


def test_the_stub_checks_the_header_s_abi_identity():
    import levelsetfusion_python_amd as pkg
    source = stub_source()
    m = re.search(r"LSF_ABI_VERSION, LSF_ABI_HASH = (\d+), b\"([0-9a-f]{16})\"", source)
    assert m, "the stub names the ABI version and header hash it was written against"
    assert int(m.group(1)) == pkg._lib.ABI_VERSION
    assert m.group(2) == pkg._lib.HEADER_ABI_HASH == pkg._build.abi_hash()
    # ... and refuses a library of another identity before its first compute call
    check = source.index("lib.lsf_abi_hash() != LSF_ABI_HASH")
    assert check < source.index("def warp_field_advanced")
    assert re.search(r"#define\s+LSF_ABI_VERSION\s+%s\b" % m.group(1), open(HEADER).read())


def test_the_stub_refuses_a_library_of_another_abi(tmp_path):
    """the stub's own check, run against the real library with a doctored expectation (no GPU needed: lsf_abi_hash and
    lsf_abi_version are host functions)"""
    source = stub_source()
    head = source[:source.index("class lsf_grid")]
    assert "LSF_ABI_HASH" in head
    doctored = re.sub(r"b\"[0-9a-f]{16}\"", "b\"0000000000000000\"", head, count=1)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(head, {})                      # the header's identity: loads and passes
        with pytest.raises(ImportError):
            exec(doctored, {})
    finally:
        os.chdir(cwd)


@pytest.mark.gpu
def test_the_documented_stub_reproduces_the_reference_s_known_answers():
    T = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_literals.npz"))
    scope = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                          # the stub loads the library by its path relative to the repository root
    try:
        exec(stub_source(), scope)
    finally:
        os.chdir(cwd)
    warp_field_advanced = scope["warp_field_advanced"]
    flags = {"01": (False, False, False), "02": (True, False, True), "03": (False, False, False),
             "04": (False, False, False), "05": (False, False, False)}
    for case, fl in flags.items():
        p = "field_warping.test_warp_field_advanced%s." % case
        new_live, (u, v) = warp_field_advanced(T[p + "warped_live_template"].copy(), T[p + "canonical_field"],
                                               T[p + "u_vectors"].copy(), T[p + "v_vectors"].copy(), *fl)
        assert np.allclose(new_live, T[p + "expected_new_warped_live_field"], atol=1e-5), case   # north-star tolerance
        if p + "expected_u_vectors" in T.files:
            assert np.allclose(u, T[p + "expected_u_vectors"], atol=1e-5), case
            assert np.allclose(v, T[p + "expected_v_vectors"], atol=1e-5), case
