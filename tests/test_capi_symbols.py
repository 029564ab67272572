"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/jrlqp_hip.h declares; argument validation returns error codes (no compute is launched)."""
import ctypes as ct
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jrlqp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jrlqp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    import jrl_qp_amd.capi as capi

    lib = capi.load_library()
    decl = declared_symbols()
    assert decl, "no declarations found in the header"
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/jrlqp_hip.h but not exported"
    assert sorted(capi.EXPORTED) == decl
    assert lib.jrlqp_abi_version() == 4


def test_defaults_and_lds_budget(pkg):
    import jrl_qp_amd.capi as capi

    lib = capi.load_library()
    o = capi.Options()
    lib.jrlqp_default_options(ct.byref(o))
    assert o.max_iter == 500 and o.big_bnd == 1e100  # reference include/jrl-qp/SolverOptions.h:16-17
    c = ct.c_int(-1)
    lds = lib.jrlqp_gi_batch_lds_bytes(50, 200, 0, ct.byref(c))
    assert 0 < lds <= 163840 // 3 and c.value == 2  # BASELINE config c3: C in registers, three workgroups per CU
    lds = lib.jrlqp_gi_batch_lds_bytes(10, 20, 0, ct.byref(c))
    assert 0 < lds <= 163840 // 5 and c.value == 2  # c1 shape: its own register-resident instantiation, five workgroups per CU
    l12 = lib.jrlqp_gi_batch_lds_bytes(12, 24, 0, ct.byref(c))
    l16 = lib.jrlqp_gi_batch_lds_bytes(16, 24, 0, ct.byref(c))
    assert l12 == l16 and c.value == 2  # a size without an instantiation runs embedded in the next one (cold solves)
    lds = lib.jrlqp_gi_batch_lds_bytes(5, 10, 0, ct.byref(c))
    assert 0 < lds < 16384 and c.value == 1  # tiny problems: the generic kernel, C staged in LDS
    w12 = lib.jrlqp_gi_batch_warm_lds_bytes(12, 24, 0, ct.byref(c))
    w16 = lib.jrlqp_gi_batch_warm_lds_bytes(16, 24, 0, ct.byref(c))
    assert w12 == w16 and c.value == 2  # warm starts are embedded the same way
    lds = lib.jrlqp_gi_batch_lds_bytes(50, 2000, 50, ct.byref(c))
    assert 0 < lds <= 163840 and c.value == 0  # too many constraints: C is streamed
    lds = lib.jrlqp_gi_batch_lds_bytes(100, 200, 0, ct.byref(c))
    assert 100 * 101 * 8 * 1.5 < lds <= 163840 and c.value == 0  # 64 < n <= 128: J and the packed R in LDS, C streamed
    lds = lib.jrlqp_gi_batch_lds_bytes(128, 256, 128, ct.byref(c))
    assert 128 * 129 * 8 < lds <= 163840  # J alone: R goes to the scratch workspace
    assert lib.jrlqp_gi_batch_warm_lds_bytes(65, 10, 0, ct.byref(c)) > 65 * 65 * 8 and c.value == 0  # (warm starts beyond one wavefront: gi_batch_big)
    assert lib.jrlqp_gi_batch_warm_lds_bytes(129, 10, 0, ct.byref(c)) > 0 and lib.jrlqp_gi_batch_warm_lds_bytes(257, 10, 0, ct.byref(c)) == -1
    assert 0 < lib.jrlqp_gi_batch_lds_bytes(129, 10, 0, ct.byref(c)) < 40000 and c.value == 0  # 128 < n <= 256: J in the workspace, vectors in LDS
    assert lib.jrlqp_gi_batch_lds_bytes(257, 10, 0, ct.byref(c)) == -1
    assert lib.jrlqp_gi_batch_lds_bytes(10, 10, 5, ct.byref(c)) == -1  # nb must be 0 or n


def test_argument_validation_without_gpu(pkg):
    import jrl_qp_amd.capi as capi

    lib = capi.load_library()
    args = [None] * 7 + [None] + [None] * 8 + [0, None]
    assert lib.jrlqp_gi_solve_batch(0, 1, 0, 1, *args) == -1  # n < 1
    assert lib.jrlqp_gi_solve_batch(4, 1, 3, 1, *args) == -1  # nb not in {0, n}
    assert lib.jrlqp_gi_solve_batch(4, 1, 0, 0, *args) == 0  # empty batch is a no-op
    assert lib.jrlqp_gi_solve_batch(4, 1, 0, 1, *args) == -1  # null pointers
    assert lib.jrlqp_status_string(-2).decode().startswith("problem size")
    # the workspace contract (no GPU needed for the shapes that need none)
    assert lib.jrlqp_gi_batch_workspace_bytes(50, 200, 0, 4096) == 0
    assert lib.jrlqp_gi_batch_workspace_bytes(100, 200, 0, 64) != 0  # beyond one wavefront: B = L^-1 N of a warm start (-1 without a device: sized by its CU count)
    assert lib.jrlqp_gi_batch_workspace_bytes(257, 10, 0, 1) == -1 and lib.jrlqp_gi_batch_workspace_bytes(10, 10, 5, 1) == -1
    assert lib.jrlqp_gi_batch_workspace_bytes(256, 10, 0, 0) == 0  # an empty batch needs nothing
    ws_args = [None] * 8 + [None] + [None] * 8 + [0, None, 0, None]
    assert lib.jrlqp_gi_solve_batch_ws(0, 4, 1, 0, 0, *ws_args) == 0  # empty batch
    assert lib.jrlqp_gi_solve_batch_ws(0, 4, 1, 0, 1, *ws_args) == -1  # null pointers
    ws_bad = [None] * 8 + [None] + [None] * 8 + [0, None, 64, None]
    assert lib.jrlqp_gi_solve_batch_ws(0, 4, 1, 0, 1, *ws_bad) == -1  # a size without a pointer


def test_shard_arithmetic_of_the_multi_device_entry(pkg):
    """jrlqp_shard_range (the split jrlqp_gi_solve_batch_multi applies) == jrl-qp_amd/dist.py shard_range (the split of the
    one-process-per-GPU launcher): contiguous, balanced, covering, in order -- BASELINE c4 (32768 over 8) included."""
    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.dist as jdist

    for total in (0, 1, 7, 8, 9, 100, 4096, 32768, 32771):
        for world in (1, 2, 3, 4, 6, 8):
            nxt = 0
            for r in range(world):
                first, count = capi.shard_range(total, world, r)
                lo, hi = jdist.shard_range(total, r, world)
                assert (first, first + count) == (lo, hi) and first == nxt
                assert count in (total // world, total // world + 1)
                nxt = first + count
            assert nxt == total
    assert capi.shard_range(32768, 8, 3) == (3 * 4096, 4096)
    assert capi.shard_range(10, 4, 7) == (0, 0) and capi.shard_range(-1, 4, 0) == (0, 0)  # out of range: an empty shard
    lib = capi.load_library()
    assert lib.jrlqp_gi_solve_batch_multi(0, 4, 1, 0, 8, 0, None, None, None, 1) == -1  # no device list
    ids = (ct.c_int * 1)(0)
    assert lib.jrlqp_gi_solve_batch_multi(0, 4, 1, 0, 8, 1, ids, None, None, 1) == -1  # no shard buffers
    assert lib.jrlqp_gi_gather_x_multi(4, 8, 1, ids, None, 0, None, 1) == -1


def test_blockgi_plan_validation_without_gpu(pkg):
    """jrlqp_blockgi_create rejects inconsistent shapes before touching the device (no GPU needed)."""
    import jrl_qp_amd.capi as capi

    lib = capi.load_library()
    h = ct.c_void_p()
    I = lambda *v: (ct.c_int * len(v))(*v)
    # StructuredC must cover exactly the unknowns of G (BlockGISolver.cpp:35)
    assert lib.jrlqp_blockgi_create(2, 2, I(3, 4), 2, I(3, 3), I(2, 2), 0, 1, 100, ct.byref(h)) == -1
    assert lib.jrlqp_blockgi_create(5, 2, I(3, 4), 2, I(3, 4), I(2, 2), 0, 1, 100, ct.byref(h)) == -1  # kind
    assert lib.jrlqp_blockgi_create(2, 2, I(3, 4), 2, I(3, 4), I(2, 2), 0, 0, 100, ct.byref(h)) == -1  # nprob
    assert lib.jrlqp_blockgi_create(2, 2, I(3, 4), 2, I(3, 4), I(2, 2), 0, 1, 0, ct.byref(h)) == -1  # max_iter
    assert lib.jrlqp_blockgi_create(2, 2, I(3, 130), 2, I(3, 130), I(2, 2), 0, 1, 10, ct.byref(h)) == -2  # block > 128 rows
    assert lib.jrlqp_blockgi_create(2, 70, I(*([60] * 70)), 1, I(4200), I(1), 0, 1, 10, ct.byref(h)) == -2  # n > 4096
    assert lib.jrlqp_blockgi_workspace_bytes(None) == -1
    assert lib.jrlqp_blockgi_solve(None, *([None] * 12), None, 0, None) == -1
    # ABI 4: the form of the QR factors is AUTO (-1), SEQUENCE (0) or C9 (1); anything else is refused before the device is touched,
    # and so are the same inconsistent shapes through the new entry point
    for bad in (-2, 2, 7):
        assert lib.jrlqp_blockgi_create_form(0, 2, I(3, 4), 2, I(3, 4), I(2, 2), 0, 1, 100, bad, ct.byref(h)) == -1
    assert lib.jrlqp_blockgi_create_form(0, 2, I(3, 4), 2, I(3, 3), I(2, 2), 0, 1, 100, 1, ct.byref(h)) == -1
    assert lib.jrlqp_blockgi_create_form(0, 2, I(3, 130), 2, I(3, 130), I(2, 2), 0, 1, 10, 1, ct.byref(h)) == -2
    assert lib.jrlqp_blockgi_form(None) == -1
    # the multi-device debug switches answer without a device
    assert lib.jrlqp_debug_multi_allow_shared_device(0) == 0 and lib.jrlqp_debug_multi_inject_failure(-1) == 0
    assert lib.jrlqp_gi_multi_failed_shard() == -1


def test_product_does_not_touch_the_oracle():
    """The product path must not import, link or execute anything under oracle/ (only tests/, smoke() and
    bench.py's cpu_baseline may)."""
    pk = os.path.join(ROOT, "jrl-qp_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in text and "libjrlqp_oracle" not in text and "jrlqp_oracle.h" not in text, f
    text = open(os.path.join(ROOT, "include", "jrlqp_hip.h")).read()
    assert "jrlqp_oracle" not in text
