"""CPU-side checks of the C ABI: the library loads, exports every symbol the
header declares, the ctypes table covers them all, and the HOST entry point
(iif_build_table) agrees with the golden tables.  No device work."""
import ctypes
import os
import re

import numpy as np
import pytest

from iif_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "iif_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(iif_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = header_symbols()
    assert "iif_ce_fwd_bwd" in names and "iif_version" in names
    for n in names:
        assert hasattr(lib, n), "libiif_amd.so does not export %s" % n
    assert lib.iif_version().decode().startswith("iif_amd")


def test_ctypes_table_covers_header():
    assert sorted(list(_lib.SIGNATURES) + ["iif_version"]) == header_symbols()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libiif_amd.so")
    with pytest.raises(_lib.IIFNativeError):
        _lib.lib()


def test_cpu_tensor_is_rejected_not_emulated():
    import torch
    from iif_amd.custom import IIFLoss

    class DS:
        def get_cls_num_list(self):
            return [500, 100, 20, 5]
    crit = IIFLoss(DS(), device="cpu")
    with pytest.raises(_lib.IIFNativeError):
        crit(torch.randn(2, 4), torch.tensor([0, 1]))


@pytest.mark.parametrize("name", ["c4", "cifar100_exp100", "places365", "imagenet1000", "lvis1204"])
def test_host_build_table_matches_golden(golden, name):
    g = golden("g3_tables")
    counts = np.ascontiguousarray(g[name + "_counts"], dtype=np.int64)
    C = len(counts)
    lib = _lib.lib()
    worst = 0
    for norm in (0, 1, 2):
        for v, code in _lib.VARIANT_CODE.items():
            out = np.empty(C, dtype=np.float32)
            rc = lib.iif_build_table(counts.ctypes.data, C, code, norm, out.ctypes.data)
            assert rc == 0
            ref = g["%s_n%d_%s" % (name, norm, v)].reshape(-1)
            ulp = np.abs(out.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
            if norm == 0:
                # libm vs numpy/scipy differ by at most the last float32 bit
                worst = max(worst, int(ulp))
                assert ulp <= 1, (name, norm, v, int(ulp))
            else:
                # the divisor is torch's float32 p-norm (custom.py:25-26); its CPU accumulation
                # order is not reproduced in C (double accumulation here) -> relative 2e-6
                np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6 * float(np.abs(ref).max()))
    assert worst <= 1


def test_host_build_table_rejects_bad_arguments():
    lib = _lib.lib()
    out = np.empty(4, dtype=np.float32)
    counts = np.array([1, 2, 3, 4], dtype=np.int64)
    assert lib.iif_build_table(0, 4, 0, 0, out.ctypes.data) == -1
    assert lib.iif_build_table(counts.ctypes.data, 0, 0, 0, out.ctypes.data) == -1
    assert lib.iif_build_table(counts.ctypes.data, 4, 9, 0, out.ctypes.data) == -1


def test_routing_queries_and_argument_checks_of_the_round6_entries():
    """Host logic of the late round-6 entry points, no device needed: which (cs, cd, c2) the producer with the P / Gram
    by-product takes (one N slice of 256 columns, a 64-channel second source, at least 1 024 rows in whole 64-row tiles), and
    the argument checks of iif_slab_sum (they return before anything is launched)."""
    import torch
    from iif_amd import ops
    ok = lambda n, hw, cs, cd, c2: ops.conv_dgrad_rx_pg_ok(n, hw, hw, cs, cd, c2, torch.bfloat16)   # noqa: E731
    assert ok(256, 56, 64, 256, 64) and ok(256, 56, 128, 256, 64) and ok(4, 16, 64, 256, 64)
    assert not ok(256, 28, 128, 512, 128)        # two N slices: no block holds every column of g~
    assert not ok(256, 56, 64, 256, 128)         # 128-channel second source
    assert not ok(256, 14, 256, 1024, 256)       # no recomputing instance at all
    assert not ok(1, 16, 64, 256, 64)            # 256 rows: below the persistent kernel's minimum
    assert not ops.conv_dgrad_rx_pg_ok(256, 56, 56, 64, 256, 64, torch.float32)
    L = _lib.lib()
    raw = (ctypes.c_float * 80)()
    addr = (ctypes.addressof(raw) + 15) // 16 * 16            # the entry wants 16-byte aligned buffers
    assert L.iif_slab_sum(None, 64, 1, 4, 4, 4, addr, None) == -1
    assert L.iif_slab_sum(addr, 64, 0, 4, 4, 4, addr, None) == -1           # no slab
    assert L.iif_slab_sum(addr, 64, 1, 4, 6, 4, addr, None) == -1           # pitch not a multiple of 4
    assert L.iif_slab_sum(addr, 64, 1, 4, 4, 8, addr, None) == -1           # more columns than the pitch
    assert L.iif_slab_sum(addr, 15, 1, 4, 4, 4, addr, None) == -1           # slabs beyond the buffer


def test_python_tables_bit_exact_with_golden(golden):
    import torch
    from iif_amd.custom import build_tables
    g = golden("g3_tables")
    for name in ("c4", "cifar100_exp100", "imagenet1000", "lvis1204"):
        counts = g[name + "_counts"].tolist()
        for norm in (0, 1, 2):
            tabs = build_tables(counts, norm)
            for v in tabs:
                assert torch.equal(tabs[v], torch.from_numpy(g["%s_n%d_%s" % (name, norm, v)])), (name, norm, v)


def test_c_oracle_matches_golden(golden):
    """The plain-C restatement (oracle/iif_oracle.c) against the reference's vectors."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libiif_oracle.so"))
    lib.iif_oracle_ce.restype = ctypes.c_double
    P, D, I, L = ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int64
    lib.iif_oracle_ce.argtypes = [P, P, P, P, D, P, P, L, D, I, I, P, P]
    g, t = golden("g4_loss"), golden("g3_tables")
    for name in ("c4", "cifar100_exp100", "imagenet1000"):
        pred = np.ascontiguousarray(g[name + "_pred"]); tgt = np.ascontiguousarray(g[name + "_targets"])
        cw = np.ascontiguousarray(g[name + "_class_weight"], dtype=np.float32)
        B, C = pred.shape
        for v in ("raw", "gombit"):
            tab = np.ascontiguousarray(t["%s_n0_%s" % (name, v)].reshape(-1))
            for red, scale in (("mean", 1.0 / B), ("sum", 1.0)):
                for wname, w in (("nw", None), ("cw", cw)):
                    rows = np.empty(B); d = np.empty((B, C))
                    loss = lib.iif_oracle_ce(pred.ctypes.data, tab.ctypes.data, tgt.ctypes.data, None, 1.0, None,
                                             None if w is None else w.ctypes.data, -100, scale, B, C,
                                             rows.ctypes.data, d.ctypes.data)
                    key = "%s_%s_%s_%s" % (name, v, red, wname)
                    ref_l, ref_d = float(g[key + "_loss"]), g[key + "_dpred"]
                    assert abs(loss - ref_l) <= 2e-6 * max(1.0, abs(ref_l)), key
                    assert np.abs(d - ref_d).max() <= 2e-6 * max(1.0, np.abs(ref_d).max()), key


def _kernel_bodies(asm, name_part):
    """{mangled name: [instruction lines]} of the functions whose mangled name contains ``name_part``."""
    out, cur = {}, None
    for line in asm.splitlines():
        t = line.strip()
        if line and not line[0].isspace() and t.startswith("_Z") and ":" in t and name_part in t.split(":")[0]:
            cur = t.split(":")[0]
            out[cur] = []
        elif cur is not None:
            if t.startswith("s_endpgm"):
                cur = None
            elif t and not t.startswith((".", ";")):
                out[cur].append(t)
    return out


@pytest.mark.parametrize("src,kernel", [("bn.hip", "bn_reduce_finalize_kernel"), ("iif_head.hip", "row_reg_kernel")])
def test_ticket_protocol_isa(tmp_path, src, kernel):
    """The single-launch reductions (BN statistics / BN-backward sums, the IIF loss) publish partials with returning
    agent-scope atomic exchanges, wait for them (`s_waitcnt vmcnt(0)`), take a ticket with an atomic add, and the last
    block reads the partials back with scope-qualified loads.  The HIP memory model does not promise that order for relaxed
    atomics; the hardware does, PROVIDED the compiler emits exactly this sequence.  This test pins the emitted gfx950 ISA:
      swap ... sc0 (returning)  ->  s_waitcnt vmcnt(0)  ->  global_atomic_add ... sc0  ->  global_load ... sc1
    (cross-compiles the one file to assembly: no GPU needed)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "iif_amd", "csrc")
    asm_path = tmp_path / "k.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + csrc, "--offload-device-only",
                    "-S", "-o", str(asm_path), os.path.join(csrc, src)], check=True, stderr=subprocess.DEVNULL)
    bodies = _kernel_bodies(asm_path.read_text(), kernel)
    assert bodies, "kernel %s not found in the assembly of %s" % (kernel, src)
    checked = 0
    for name, ins in bodies.items():
        # the publishing exchanges and the ticket are the RETURNING forms (sc0); a non-returning swap is something else
        # (the loss kernel's bad-target status flag)
        swaps = [i for i, t in enumerate(ins) if t.startswith("global_atomic_swap") and " sc0" in t]
        adds = [i for i, t in enumerate(ins) if t.startswith("global_atomic_add") and " sc0" in t]
        if not swaps and not adds:
            continue                                    # an instantiation without the ticket (e.g. no scalar reduction)
        assert swaps and adds, name
        ticket = adds[0]
        assert max(swaps) < ticket, name
        between = ins[max(swaps) + 1:ticket]
        assert any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in between), (name, between[:8])
        loads_after = [t for t in ins[ticket:] if t.startswith("global_load")]
        assert any(" sc1" in t for t in loads_after), (name, loads_after[:4])
        checked += 1
    assert checked >= 1


def _vregs(tok):
    """VGPR numbers an operand token names: v12 -> [12], v[12:15] -> [12..15]; anything else -> []."""
    import re
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


@pytest.mark.parametrize("src", ["conv_wgrad.hip", "bn3_algebra.hip", "conv_regw.hip"])
def test_tr_read_results_are_waited_for(tmp_path, src):
    """The transposing LDS reads are inline asm (the builtin drains every LDS-DMA in flight, conv_wgrad.hip:51-56), so the
    compiler does not know that their destination registers are still being written.  Pin the emitted gfx950 ISA: between a
    `ds_read_b64_tr_b16 vD, ...` and the `s_waitcnt lgkmcnt(N)` that covers it (LDS operations return in order: the wait covers
    a read once at most N LDS instructions were issued after it), NO instruction may name a register of vD - not an MFMA,
    and not a register copy the allocator placed there to assemble a 128-bit fragment."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "iif_amd", "csrc")
    asm_path = tmp_path / "k.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + csrc, "--offload-device-only",
                    "-S", "-o", str(asm_path), os.path.join(csrc, src)], check=True, stderr=subprocess.DEVNULL)
    bodies = _kernel_bodies(asm_path.read_text(), "")
    reads = 0
    for name, ins in bodies.items():
        pending = {}                 # vgpr -> index of its read among the LDS instructions issued so far
        n_lds = 0
        for t in ins:
            op, _, rest = t.partition(" ")
            toks = [x.strip() for x in re.split(r"[ ,]+", rest.split(";")[0]) if x.strip()]
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    pending = {r: i for r, i in pending.items() if i > n_lds - keep}
                continue
            named = [r for tok in toks for r in _vregs(tok)]
            if op == "ds_read_b64_tr_b16":
                dst = _vregs(toks[0])
                clash = [r for r in named[len(dst):] if r in pending]
                assert not clash, (name, t, "address register still pending")
                n_lds += 1
                reads += 1
                for r in dst:
                    assert r not in pending, (name, t, "destination rewritten while a read into it is in flight")
                    pending[r] = n_lds
                continue
            clash = [r for r in named if r in pending]
            assert not clash, "%s: `%s` touches v%s before the s_waitcnt that covers its ds_read_b64_tr_b16" % (name, t, clash)
            if op.startswith("ds_"):
                n_lds += 1
    assert reads >= 8, "no transposing reads found in %s" % src


def test_counted_vmcnt_waits_of_the_register_weight_kernels(tmp_path):
    """The persistent 1x1 kernels of conv_regw.hip close every tile with a HAND-COUNTED `s_waitcnt vmcnt(N)`: the next tile's
    LDS-DMA pieces and epilogue operands must have landed, this tile's N stores may stay in flight (vector-memory operations
    retire in order).  That is only right if the emitted loop body issues AT LEAST N vector-memory instructions behind the last
    load it waits for, all of them stores - a compiler that dropped, merged or hoisted one of them would make the wait let a
    load through (round-5 review: the bit-identity tests were the only net).  Pin the gfx950 ISA: walking the tile loop
    BACKWARDS from each hand-written counted wait (cyclically: the compiler rotates some of these loops), at least N stores come
    before the first LDS-DMA / load."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "iif_amd", "csrc")
    asm_path = tmp_path / "k.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + csrc, "--offload-device-only",
                    "-S", "-o", str(asm_path), os.path.join(csrc, "conv_regw.hip")], check=True, stderr=subprocess.DEVNULL)
    cur, hand, kernels = None, False, {}
    for line in asm_path.read_text().splitlines():
        t = line.strip()
        if line and not line[0].isspace() and t.startswith("_Z") and ": ;" in t and "gemm1x1_regw_kernel" in t.split(":")[0]:
            cur = t.split(":")[0]
            kernels[cur] = []
        elif cur is not None:
            if t.startswith(";;#ASMSTART"):
                hand = True
            elif t.startswith(";;#ASMEND"):
                hand = False
            elif t.startswith("s_endpgm"):
                cur = None
            elif t.startswith(".LBB") and t.split(";")[0].strip().endswith(":"):
                kernels[cur].append(("LABEL " + t.split(":")[0], False))
            elif t and not t.startswith((".", ";")):
                kernels[cur].append((t, hand))
    assert len(kernels) >= 15, sorted(kernels)
    checked = 0
    for name, ins in kernels.items():
        labels = {t.split()[1]: i for i, (t, _) in enumerate(ins) if t.startswith("LABEL ")}
        loops = []                                                     # (first, last) of every backward branch
        for i, (t, _) in enumerate(ins):
            m = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        for i, (t, h) in enumerate(ins):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)$", t)
            if not (h and m and int(m.group(1)) > 0):
                continue
            n = int(m.group(1))
            inside = [lp for lp in loops if lp[0] <= i <= lp[1]]
            assert inside, (name, t, "a counted wait outside the tile loop")
            lo, hi = max(inside, key=lambda lp: lp[1] - lp[0])       # (the tile loop: block placement also produces short backward jumps)
            body = ins[lo:hi + 1]
            order = body[:i - lo][::-1] + body[i - lo + 1:][::-1]      # backwards from the wait, wrapping round the loop
            stores = 0
            for (u, _) in order:
                op = u.split()[0]
                if (op.startswith("buffer_load") and u.rstrip().endswith("lds")) or op.startswith(("global_load", "scratch_load")):
                    break
                if op.startswith(("global_store", "scratch_store", "buffer_store")):
                    stores += 1
            assert stores >= n, "%s: vmcnt(%d) has only %d stores between it and the last load of the tile loop" % (name, n, stores)
            checked += 1
    assert checked >= 15, checked
