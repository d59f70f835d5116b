"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/misslap.h declares,
the ctypes structs match the header, and the front-end's host logic (argument validation, error texts)
behaves like the reference's -- without any compute call (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sslap_amd import _lib, auction_solve, from_sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "misslap.h")).read()


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = set(re.findall(r"\b(misslap_[a-z_0-9]+)\s*\(", _header()))
    assert declared, "no prototypes found"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.misslap_abi_version() == int(re.search(r"MISSLAP_ABI_VERSION (\d+)", _header()).group(1))


def test_struct_layouts_match_header():
    """Compile a tiny C program against the header and compare sizeof / offsetof with ctypes."""
    import subprocess
    import tempfile
    fields = {"misslap_options": ["max_iter", "tail_threshold", "rounds_per_sync", "tiled_min_K", "cand_mode",
                                  "cand_refresh_min", "reserved", "input_stream"],
              "misslap_meta": ["struct_size", "start_eps", "its", "obj_f64", "edges_scanned", "bid_ms", "tail_edges", "tiled_min_K", "bid_edges_read", "fullscan_edges_read", "shard_edges", "cand_hits", "tail_stats",
                               "complete_assignment", "valid_assignment", "lines_active", "sharded_rounds", "tiled_format", "phases_with_lines", "eps_phases", "filter_undecided"],
              "misslap_status": ["K", "error_bits", "rounds_per_sync", "shard_min_K"]}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "misslap.h"', 'int main(void){']
    for s, fs in fields.items():
        prog.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in fs:
            prog.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write("\n".join(prog))
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    types = {"misslap_options": _lib.Options, "misslap_meta": _lib.Meta, "misslap_status": _lib.Status}
    for s, t in types.items():
        assert int(out[s]) == C.sizeof(t), s
        for f in fields[s]:
            assert int(out[f"{s}.{f}"]) == getattr(t, f).offset, (s, f)


def test_options_struct_size_versions_are_recognised():
    """ABI 1 (88 bytes, knobs in reserved[]) and ABI 2 options pass the validation (and then fail for want of a GPU
    here, or succeed on a GPU box); any other size, a non-zero reserved word or a bad knob is MISSLAP_ERR_INVALID."""
    lib = _lib.load()
    loc = np.array([[0, 0], [1, 1]], dtype=np.int32)
    val = np.array([1.0, 2.0])

    def create(buf):
        h = C.c_void_p()
        rc = lib.misslap_create(C.byref(h), 2, loc.ctypes.data, val.ctypes.data, C.cast(C.byref(buf), C.POINTER(_lib.Options)))
        if rc == 0:
            lib.misslap_destroy(h)
        return rc, lib.misslap_last_error().decode()

    class OptionsV1(C.Structure):
        _fields_ = _lib.Options._fields_[:12] + [("reserved", C.c_int32 * 8)]
    assert C.sizeof(OptionsV1) == 88
    v1 = OptionsV1(struct_size=88, tail_threshold=-1, max_iter=10)
    v1.reserved[7] = 5000 | (9 << 24)
    rc, msg = create(v1)
    assert rc in (0, _lib.ERR_NO_DEVICE), msg
    v2 = _lib.Options(struct_size=C.sizeof(_lib.Options), tail_threshold=-1, max_iter=10)
    rc, msg = create(v2)
    assert rc in (0, _lib.ERR_NO_DEVICE), msg
    short = _lib.Options(struct_size=_lib.Options.reserved.offset + 4, tail_threshold=-1, max_iter=10)  # a shorter ABI-2 struct
    rc, msg = create(short)
    assert rc in (0, _lib.ERR_NO_DEVICE), msg
    for bad in (_lib.Options(struct_size=90), _lib.Options(struct_size=4096), _lib.Options(struct_size=0)):
        rc, msg = create(bad)
        assert rc == _lib.ERR_INVALID and "struct_size" in msg, (rc, msg)
    nz = _lib.Options(struct_size=C.sizeof(_lib.Options))
    nz.reserved[3] = 1
    rc, msg = create(nz)
    assert rc == _lib.ERR_INVALID and "reserved" in msg
    rc, msg = create(_lib.Options(struct_size=C.sizeof(_lib.Options), cand_mode=7))
    assert rc == _lib.ERR_INVALID and "cand_mode" in msg
    rc, msg = create(_lib.Options(struct_size=C.sizeof(_lib.Options), tiled_shape=99))  # beyond the shape table
    assert rc == _lib.ERR_INVALID and "tiled_shape" in msg
    rc, msg = create(_lib.Options(struct_size=C.sizeof(_lib.Options), tiled_shape=5))  # a valid shape (the column split): only the GPU is missing
    assert rc in (0, _lib.ERR_NO_DEVICE), msg


def test_cache_limits_and_comm_info_need_no_gpu():
    """misslap_set_cache_limits validates its arguments; misslap_comm_info reports a custom communicator as its ops
    describe it (for RCCL it is ncclCommCount: tests/test_gpu_parity.py::test_rccl_world1_smoke)."""
    lib = _lib.load()
    assert lib.misslap_set_cache_limits(-1, 0, 0) == _lib.ERR_INVALID
    assert lib.misslap_set_cache_limits(4 << 30, 1 << 30, 64) == 0  # the library's defaults
    from sslap_amd.dist import Comm
    c = Comm.custom(1, 2, lambda p, n, s: None, lambda p, n, s: None)
    assert c.info() == dict(kind="custom", rank=1, world=2, transport_ranks=2)
    assert lib.misslap_comm_info(None, None, None, None, None) == _lib.ERR_INVALID
    rd = C.c_double()
    assert lib.misslap_measure_hbm(0, 1 << 20, 0, C.byref(rd), None) == _lib.ERR_INVALID  # reps < 1


def test_trim_caches_needs_no_gpu():
    freed = C.c_int64(-1)
    assert _lib.load().misslap_trim_caches(C.byref(freed)) == 0 and freed.value >= 0


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU every solver call must raise, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    loc = np.array([[0, 0], [1, 1]], dtype=np.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        auction_solve(loc=loc, val=np.array([1.0, 2.0]), cardinality_check=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        auction_solve(mat=np.ones((2, 2)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sslap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("# oracle-free", ""), f"{f} mentions the oracle"


def test_frontend_validation_before_ffi():
    loc = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.int64)
    with pytest.raises(ValueError, match="One of the following formats is expected"):
        auction_solve()
    with pytest.raises(ValueError, match="Buffer dtype mismatch, expected 'DTYPE_t' but got 'float'"):
        auction_solve(loc=loc, val=np.ones(4, dtype=np.float32), cardinality_check=False)
    with pytest.raises(ValueError, match="expected 'double' but got 'float'"):
        auction_solve(mat=np.ones((2, 2), dtype=np.float32))
    with pytest.raises(ValueError, match="expected 'double' but got 'long'"):
        auction_solve(mat=np.ones((2, 2), dtype=np.int64))
    with pytest.raises(ValueError, match="wrong number of dimensions"):
        auction_solve(loc=loc, val=np.ones((4, 1)), cardinality_check=False)
    # "fewer than N entries" guard with the adapter's own N (auction_.pyx:592-595, :604)
    few = np.array([[0, 0], [5, 1]], dtype=np.int64)
    with pytest.raises(ValueError, match="Fewer than 5 valid values provided for 5 rows"):
        from_sparse(few, np.ones(2), cardinality_check=False)
    with pytest.raises(ValueError, match="Fewer than 9 valid values provided for 9 rows"):
        from_sparse(few, np.ones(2), size=(6, 9), cardinality_check=False)  # `M, N = size` (sic)
    # infeasible graph caught by the (host) cardinality guard before any GPU work
    inf_loc = np.array([[0, 0], [1, 0]], dtype=np.int64)
    with pytest.raises(ValueError, match="Maximum matching possible only involves 1 out of 2 rows"):
        auction_solve(loc=inf_loc, val=np.ones(2), size=(2, 2))


def test_header_is_plain_c_and_the_c_client_compiles(tmp_path):
    """include/misslap.h must be consumable by a C compiler (no C++-isms, no torch / HIP types)."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([cc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           "-c", os.path.join(root, "tests", "cabi_client.c"), "-o", str(tmp_path / "client.o")])


def test_rccl_binding_matches_the_installed_header():
    """The RCCL path with more than one rank cannot run on a one-GPU box; what CAN be pinned without a GPU: the library's
    own dlopen of librccl finds a copy, every entry point the exchange uses resolves, and the enumerator values and the
    id size it compiled in equal the installed rccl.h (a silent change of ncclDataType_t / ncclRedOp_t would turn the
    MAX all-reduce into something else on the first 8-GPU run)."""
    hdr = None
    for cand in ("/opt/rocm/include/rccl/rccl.h", "/opt/rocm/include/rccl.h"):
        if os.path.exists(cand):
            hdr = open(cand).read()
            break
    if hdr is None:
        pytest.skip("rccl.h is not installed")
    n, enums, path = C.c_int32(), (C.c_int32 * 6)(), C.create_string_buffer(512)
    rc = _lib.load().misslap_rccl_selfcheck(C.byref(n), enums, path, 512)
    assert rc == 0, _lib.load().misslap_last_error()
    assert n.value == 6 and b"rccl" in path.value

    def enum_value(name):
        m = re.search(r"\b%s\s*=\s*(\d+)" % name, hdr)
        assert m, name
        return int(m.group(1))
    assert [enums[k] for k in range(4)] == [enum_value("ncclInt32"), enum_value("ncclInt64"), enum_value("ncclMax"),
                                            enum_value("ncclMin")]
    assert enums[4] == int(re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+(\d+)", hdr).group(1)) == 128
    assert enums[5] >= 20000  # ncclGetVersion of the copy the library bound to (2.x.y as 2xxyy)
    # the prototypes the function pointers were declared from (argument order of the two calls that carry data)
    assert re.search(r"ncclAllReduce\(const void\*\s*sendbuff,\s*void\*\s*recvbuff,\s*size_t count,\s*ncclDataType_t datatype,\s*"
                     r"ncclRedOp_t op,\s*ncclComm_t comm,\s*hipStream_t stream\)", hdr)
    assert re.search(r"ncclCommInitRank\(ncclComm_t\*\s*comm,\s*int nranks,\s*ncclUniqueId commId,\s*int rank\)", hdr)
    assert re.search(r"ncclCommCount\(const ncclComm_t comm,\s*int\*\s*count\)", hdr)


def test_solve_batch_rejects_bad_arguments_without_a_gpu():
    lib = _lib.load()
    assert lib.misslap_solve_batch(None, 0, None, None, 0, None) == _lib.ERR_INVALID
    assert lib.misslap_solve_batch(None, 3, None, None, 0, None) == _lib.ERR_INVALID
