"""CPU tests of the C-ABI library: it loads and exports every symbol
include/sbgpu.h declares; without a GPU it fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from strawberry_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from strawberry_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sbgpu.h")).read()
    declared = set(re.findall(r"\b(sbgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_constants(lib):
    assert b"gfx950" in lib.sbgpu_version()
    hdr = open(os.path.join(ROOT, "include", "sbgpu.h")).read()
    assert "#define SBGPU_EM_MAX_ITER 1000" in hdr          # include/estimate.hpp:237
    assert "#define SBGPU_EM_THETA_CHANGE_LIMIT 1e-2" in hdr  # include/estimate.hpp:241


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.sbgpu_init(0, C.byref(h))
    assert rc < 0 and not h.value
    assert b"device" in lib.sbgpu_last_error().lower()
    from strawberry_amd import em, synth
    with pytest.raises(Exception):
        em.EmBatchSolver(synth.make_random(4))
    with pytest.raises(Exception):
        em.EmSolver().init(2, [1, 2], [[.1, .2], [.3, .1]])


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, "strawberry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)
                assert "liboracle" not in src and "em_oracle" not in src, os.path.join(dirpath, f)


def test_synth_shapes_and_determinism():
    from strawberry_amd import synth
    a = synth.make_c2(n_loci=50)
    b = synth.make_c2(n_loci=50)
    np.testing.assert_array_equal(a.F, b.F)
    np.testing.assert_array_equal(a.count, b.count)
    assert (a.nrow == 32).all() and (a.niso == 8).all()
    sums = np.add.reduceat(a.count.astype(np.int64), a.row_off[:-1])
    assert (sums == 1000).all()
    # >= 1 compatible isoform per bin and >= 1 bin per isoform
    for l in range(a.n_loci):
        n, F = a.locus(l)
        assert (F > 0).any(axis=1).all() and (F > 0).any(axis=0).all()
    assert a.algorithmic_bytes() == 50 * (32 * 8 * 8 + 32 * 4 + 8 * 8 + 24)
    u = synth.make_c2(n_loci=5, unbinned=True)
    assert (u.nrow == 1000).all() and (u.count == 1).all()
    c3 = synth.make_c3(n_loci=300, total_frags=1e6)
    assert c3.n_loci == 300 and c3.niso.min() >= 1 and c3.nrow.max() <= 2000
    assert abs(c3.n_frags - 1e6) / 1e6 < 0.01


def test_header_is_plain_c_and_links(tmp_path, lib):
    """include/sbgpu.h compiles as C99 (no C++/HIP/torch types in the signatures) and a C
    program calling every entry point links against libsbgpu.so."""
    import subprocess
    from strawberry_amd import _lib
    src = tmp_path / "abi.c"
    calls = "\n".join("   p[%d] = (fn_t)%s;" % (i, name) for i, name in enumerate(_lib.SYMBOLS))
    src.write_text('#include "sbgpu.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\nint main(void) {\n   fn_t p[%d];\n%s\n'
                   '   printf("%%s %%d\\n", sbgpu_version(), (int)(sizeof(p) / sizeof(p[0])));\n   return p[0] == 0;\n}\n'
                   % (len(_lib.SYMBOLS), calls))
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lsbgpu", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "gfx950" in out.stdout, (out.stdout, out.stderr)


def test_host_entry_points_reject_bad_arguments(lib):
    """The host-only entry points (no GPU needed) validate their arguments and say why."""
    from strawberry_amd import _lib
    SBGPU_EINVAL = -1
    assert lib.sbgpu_bins_create(None, None, None, 1, 1, None, None, None) == SBGPU_EINVAL
    assert b"null" in lib.sbgpu_last_error()
    assert lib.sbgpu_bins_info(None, None) == SBGPU_EINVAL
    assert lib.sbgpu_format_context_row(None, 0, None, 0, None, 0, 0, None, None, None, None, 0, None, None, 0) == SBGPU_EINVAL
    out = (C.c_uint8 * 4)()
    assert lib.sbgpu_hit_features(-1, None, None, None, 0, None, None, None, out, None, None) == SBGPU_EINVAL
    # an exon with right < left is refused by the segment builder
    iso_off, exon_off = np.array([0, 1], np.int64), np.array([0, 1], np.int64)
    xl, xr = np.array([100], np.uint32), np.array([50], np.uint32)
    seg_off = np.zeros(2, np.int64)
    assert lib.sbgpu_segments_host(1, iso_off.ctypes.data, exon_off.ctypes.data, xl.ctypes.data, xr.ctypes.data,
                                   seg_off.ctypes.data, None, None, 0) == SBGPU_EINVAL
    # without a GPU the kernel-backed forms fail before touching anything
    import torch
    if not torch.cuda.is_available():
        an, ht = _lib.sbgpu_annotation_t(), _lib.sbgpu_hits_t()
        assert lib.sbgpu_exonbin_host(None, C.byref(an), C.byref(ht), 1, 1, None, None) == SBGPU_EINVAL


def test_bins_are_independent_of_host_thread_count(monkeypatch):
    """sbgpu_bins_create runs loci on host threads; the result must not depend on how many."""
    from oracle import OracleLib
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    loci = synth.make_gene_models(60, seed=14)
    hl, pairs = synth.make_fragments(loci, 40, seed=15, noise=0.3)
    feats = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
    feats = [(l, f) for l, f in feats if f is not None]
    rng = np.random.default_rng(3)
    annot = eb.Annotation(loci)
    hits = eb.Hits([l for l, _ in feats], [f for _, f in feats], mass=rng.choice([1.0, 0.5, 1 / 3, 2.0, 1.5], len(feats)))
    compat, key = OracleLib().exonbin_batch(annot, hits)
    ref = None
    for nt in ("1", "3", "16"):
        monkeypatch.setenv("SBGPU_HOST_THREADS", nt)
        b = eb.LocusBins(annot, hits, compat, key)
        got = [b.row_off, b.f_off, b.count, b.bin_key, b.bin_compat, b.hit_bin, b.pair_seg_off, b.pair_seg_lens,
               b.pair_implicit_mask, b.pair_iso_len, b.pair_out_index]
        if ref is None:
            ref = got
            assert b.n_bins > 300 and b.n_pairs > 500
        else:
            for x, y in zip(ref, got):
                np.testing.assert_array_equal(x, y)
