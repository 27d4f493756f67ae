"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n.startswith("mpn_") or n in ("NMS", "bbox_vote")))


def test_libmpn_exports_every_declared_symbol():
    import multipathnet_amd
    lib = multipathnet_amd.load()
    names = _declared("mpn.h")
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libmpn_hip.so does not export %s" % n
    assert lib.mpn_version() >= 200
    assert abs(lib.mpn_pick_scale(600, 1000, 600.0, 1000.0) - 1.0) == 0.0


def test_product_library_has_no_test_hooks():
    """timing-experiment code and mpn_debug_* switches must not ship.  The product .so exports none of them
    (its knobs are compile-time constants); they live in libmpn_hip_dbg.so (-DMPN_DEBUG_HOOKS), which exports the same ABI
    plus the hooks."""
    so = os.path.join(ROOT, "multipathnet_amd", "libmpn_hip.so")
    dbg = os.path.join(ROOT, "multipathnet_amd", "libmpn_hip_dbg.so")
    sym = subprocess.check_output(["nm", "--defined-only", so]).decode()  # the full symbol table: kernels are not exported (mpn.map)
    assert "mpn_debug" not in sym
    # conv3x3_wino_kernel<ABL, TC>: only ABL = 0 (no timing-experiment switches) ships; TC = 8 / 16 are the two block geometries
    assert not re.search(r"conv3x3_wino_kernelILi[1-9]", sym), "ablation instantiations of the Winograd kernel in the product build"
    assert re.search(r"conv3x3_wino_kernelILi0ELi8E", sym) and re.search(r"conv3x3_wino_kernelILi0ELi16E", sym)
    dsym = subprocess.check_output(["nm", "-D", "--defined-only", dbg]).decode()
    assert "mpn_debug_set_conv_variant" in dsym and "mpn_debug_set_nms_force_exact" in dsym
    for n in _declared("mpn.h"):
        assert re.search(r" T %s\b" % n, dsym), n
    # no process-global device scratch: the only device-pointer statics allowed are inside the registry (common.hip)
    for f in ("dense.hip", "nms.hip", "resnet.hip", "boxes.hip", "pipeline.hip"):
        txt = open(os.path.join(ROOT, "multipathnet_amd", "csrc", f)).read()
        assert not re.search(r"^\s*static\s+(float|char|void)\s*\*\s*\w+\s*=\s*nullptr", txt, flags=re.M), f


def test_libraries_export_the_header_and_nothing_else():
    """`nm -D` of either flavour shows the C ABI (mpn_*) only — no mangled mpn:: internals, no kernel stubs
    (csrc/mpn.map).  A Lua host that also loads another HIP library cannot collide with ours."""
    declared = set(_declared("mpn.h"))
    for name, hooks in (("libmpn_hip.so", False), ("libmpn_hip_dbg.so", True)):
        out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "multipathnet_amd", name)]).decode()
        syms = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
        assert syms, name
        assert not [x for x in syms if x.startswith("_Z")], (name, [x for x in syms if x.startswith("_Z")][:5])
        extra = sorted(x for x in syms if x not in declared and not (hooks and x.startswith("mpn_debug_")))
        assert not extra, (name, extra[:8])
        assert declared <= set(syms), (name, sorted(declared - set(syms))[:8])


def test_libnms_dropin_exports():
    so = os.path.join(ROOT, "multipathnet_amd", "libnms.so")
    assert os.path.exists(so), "libnms.so (drop-in for the reference's nms.c) was not built"
    out = subprocess.check_output(["nm", "-D", so]).decode()
    for n in _declared("mpn_libnms.h"):
        assert re.search(r" T %s\b" % n, out), n
    # the TH helpers stay undefined: they resolve to libTH inside a Torch7 process
    assert re.search(r" U THFloatTensor_resize2d\b", out)


def test_arg_errors_do_not_need_a_gpu():
    import multipathnet_amd
    lib = multipathnet_amd.load()
    lib.mpn_last_error.restype = ctypes.c_char_p
    rc = lib.mpn_nms_batched(None, None, 3, 10, ctypes.c_float(0.3), None, None, None, None)
    assert rc == -1 and b"invalid argument" in lib.mpn_last_error()
    rc = lib.mpn_nms_batched(None, None, 1, 1 << 20, ctypes.c_float(0.3), None, None, None, None)
    assert rc == -1
    # round 6: the ROI-pooling bin rule is validated before anything touches the device (N = 0 with a valid rule is a no-op)
    assert lib.mpn_roi_pool_forward_rule(None, 1, 8, 4, 4, None, 0, 7, 7, ctypes.c_float(1.0), ctypes.c_float(1.0), 0, 5, None, None, None) == -1
    assert b"invalid argument" in lib.mpn_last_error()
    assert lib.mpn_roi_pool_forward_rule(None, 1, 8, 4, 4, None, 0, 7, 7, ctypes.c_float(1.0), ctypes.c_float(1.0), 0, 1, None, None, None) == 0
    assert lib.mpn_version() == 600
    # the three pipeline constructors reject missing descriptors before touching the device
    h = ctypes.c_void_p()
    assert lib.mpn_frcnn_create(None, None, None, None, None, None, None, None, None, None, None, ctypes.byref(h)) == -1
    assert lib.mpn_mpnet_create(None, None, None, None, None, None, None, None, ctypes.byref(h)) == -1
    assert lib.mpn_resnet_create(None, None, None, None, None, None, ctypes.byref(h)) == -1
    assert not h.value


def test_no_oracle_on_product_path():
    """The product must never import / link the oracle (it would void every parity claim)."""
    pkg = os.path.join(ROOT, "multipathnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "mpn_oracle" not in txt and "libnms_ref" not in txt, os.path.join(dirpath, f)
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)
