"""CPU-side checks of the drop-in boundary: the C-ABI libraries load and export every symbol that
include/ssf.h declares; defaults equal the reference's initialize() defaults
(core/include/supersurfel_fusion/supersurfel_fusion.hpp:46-74); the product fails loudly without
a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from supersurfel_fusion_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for hdr in ("ssf.h", "ssf_testing.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(ssf_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_and_binding_agree():
    hdr = [s for s in declared_symbols() if not s.startswith("ssf_dbg_")]
    assert sorted(binding.ABI_SYMBOLS) == hdr


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_library_exports_every_declared_symbol(which, product_lib, oracle_lib):
    lib = product_lib if which == "product" else oracle_lib
    for sym in declared_symbols():
        assert hasattr(lib.lib, sym), "%s does not export %s" % (lib.path, sym)
    assert lib.lib.ssf_abi_version() == 3
    assert lib.backend == ("hip-gfx950" if which == "product" else "cpu-oracle")


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_default_config_is_the_reference_default(which, product_lib, oracle_lib):
    lib = product_lib if which == "product" else oracle_lib
    c = lib.default_config()
    ref = dict(cell_size=16, lambda_pos=50.0, lambda_bound=1000.0, lambda_size=10000.0, lambda_disp=1e6,
               seg_iter=10, seg_use_ransac=1, nb_samples=16, filter_iter=4, filter_beta=1.0, range_max=5.0,
               delta_t=20, conf_thresh=2500.0, nb_supersurfels_max=50000, icp_iter=10, icp_cov_thresh=0.04,
               rng_seed=1234, nranks=1)
    for k, v in ref.items():
        assert getattr(c, k) == pytest.approx(v), k
    assert c.thresh_disp == pytest.approx(1e-4) and c.filter_alpha == pytest.approx(0.1)
    assert c.filter_threshold == pytest.approx(0.05) and c.range_min == pytest.approx(0.2)


def test_invalid_config_is_rejected(oracle_lib, product_lib):
    for lib in (oracle_lib, product_lib):
        for bad in (dict(width=0), dict(cell_size=0), dict(nb_supersurfels_max=10), dict(nranks=2, rank=2)):
            with pytest.raises(binding.SsfError):
                binding.Fusion(lib, lib.default_config(**bad))


@pytest.mark.skipif(torch.cuda.is_available(), reason="box has a GPU")
def test_product_has_no_cpu_fallback(product_lib):
    with pytest.raises(binding.SsfError, match="no HIP device"):
        binding.Fusion(product_lib, product_lib.default_config())


def test_product_never_links_or_loads_the_oracle():
    """The product sources must not reference oracle/ (a product path through the checker would
    void every parity claim)."""
    pkg = os.path.join(ROOT, "supersurfel_fusion_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, fn)).read()
                assert "libssf_oracle" not in txt and "oracle/" not in txt and "oracle_" not in txt, fn


def test_the_product_library_reads_no_environment_variable(product_lib):
    """every measurement switch (SSF_...) lives in the lab build of the sources (csrc/variants/lab, -DSSF_EXPERIMENTS): the
    library a node links holds no such name -- and so no getenv of one"""
    import subprocess
    out = subprocess.run(["strings", product_lib.path], stdout=subprocess.PIPE, text=True).stdout.splitlines()
    assert [l for l in out if l.startswith("SSF_")] == []
    for src in ("ssf_extract.hip", "ssf_pass_tile.hpp", "ssf_track_fuse.hip", "ssf_tile_rows.inc", "ssf_host.hip", "ssf_device.hpp", "ssf_math.hpp"):
        txt = open(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", src)).read()
        body = txt.split("#ifdef SSF_EXPERIMENTS\n#include <stdlib.h>")[0] if src == "ssf_device.hpp" else txt
        assert "getenv(" not in body, src


def test_the_product_exports_no_probe_entry_points(product_lib):
    """round 5: the ablation timers, fault injection and record dumps (ssf_dbg_time_pass / _time_icp / _extract_only /
    _stall_before_match_us / ...) exist only in the lab build; what the product exports under ssf_dbg_ is exactly the
    documented test hooks of include/ssf_testing.h (host arithmetic, no device code), and every kernel probe bit
    (SSF_PROBE) is the constant false there"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", product_lib.path], stdout=subprocess.PIPE, text=True).stdout
    exported = sorted(set(re.findall(r"\b(ssf_dbg_[a-z0-9_]+)\b", out)))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ssf_testing.h")).read(), flags=re.S)
    hooks = sorted(set(re.findall(r"\b(ssf_dbg_[a-z0-9_]+)\s*\(", txt)))
    assert exported == hooks, sorted(set(exported) ^ set(hooks))
    dev = open(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", "ssf_device.hpp")).read()
    assert "#define SSF_PROBE(dbg, bits) (false)" in dev
    for src in ("ssf_extract.hip", "ssf_track_fuse.hip"):
        body = open(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", src)).read()
        assert not re.search(r"\bdbg\s*&\s*\d", body), src          # a probe bit tested outside SSF_PROBE


def test_comm_info_without_an_exchange(oracle_lib):
    f = binding.Fusion(oracle_lib, oracle_lib.default_config(width=160, height=128, fx=131.25, fy=131.25, cx=79.5, cy=63.5, nb_supersurfels_max=2048))
    assert f.comm_info() == dict(backend="none", ranks=1, rank=0)


def test_stream_copy_rate_refuses_sizes_its_forms_would_overrun(product_lib):
    """advisor, round 5: the grid-stride forms of the copy move whole rounds of 128 / 256 MiB with an unguarded first round; sizes
    below 256 MiB (and reps < 1) are refused before any device call -- so this runs without a GPU"""
    import ctypes
    f = product_lib.lib.ssf_stream_copy_rate
    f.restype = ctypes.c_double; f.argtypes = [ctypes.c_int, ctypes.c_int]
    for mib, reps in ((16, 3), (32, 3), (128, 3), (255, 3), (1024, 0)):
        assert f(mib, reps) == -1.0, (mib, reps)
