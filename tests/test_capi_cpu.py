"""CPU tier: the C-ABI library loads without a GPU and exports every symbol include/svsdct.h
declares; host-side logic of the package (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from testlib import REPO
from svsdct import batch, native, synth


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "svsdct.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 20
    lib = native.load()
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(native.SIGNATURES) == names          # the binding covers the header, nothing more
    assert lib.svs_abi_version() == native.ABI_VERSION == 4


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("svs_"))


def test_library_exports_exactly_the_declared_symbols():
    """the product library's svs_* exports ARE the header - no measurement hook, no process-global state
    behind one (round 4 shipped svs_guard_counter_set / svs_ref_copy_dev / svs_ref_read_dev / svs_probe_cvt_pk_u8).  Those live
    in the experiments library, which is the product ABI plus exactly them."""
    from testlib import EXP_LIB_PATH, EXPERIMENT_HOOKS
    assert _exported(native.LIB_PATH) == _declared_symbols()
    assert _exported(EXP_LIB_PATH) == sorted(_declared_symbols() + list(EXPERIMENT_HOOKS))
    src = open(os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd", "csrc", "svs_capi.hip")).read()
    outside = re.sub(r"#if defined\(SVS_EXPERIMENTS\).*?#e(?:lse|ndif)", "", src, flags=re.S)
    assert "g_guard_counter" not in outside            # the kernels of the product build take no counter


def test_one_default_mode_everywhere():
    """ONE default transform mode - svsdct.batch.DEFAULT_MODE = "guarded" - and every layer resolves an
    unspecified mode through batch.resolve_mode: the NumPy-level and device-level entry points, FramePipeline, the drop-in
    operator, both drop-in video loops, bench.py.  (Round 4: host_level_mode() returned "exact", so the video loops ran the
    lane-per-block kernel while bench.py and the docs said guarded.)"""
    import ast
    import inspect
    assert batch.DEFAULT_MODE == "guarded" and batch.resolve_mode(None) == "guarded" == batch.host_level_mode()
    assert batch.mode_flags(None) == batch.mode_flags("guarded") == native.SVS_EXACT_GUARDED
    assert batch.mode_flags("fast") == 0 and batch.mode_flags("exact") == native.SVS_EXACT_POCKETFFT
    with pytest.raises(ValueError):
        batch.resolve_mode("quick")
    # no entry point carries a default of its own: every `mode` parameter defaults to None ...
    for name in ("embed_frames", "extract_frames", "embed_device", "extract_device", "embed_bgr_device", "embed_bgr_frames"):
        assert inspect.signature(getattr(batch, name)).parameters["mode"].default is None, name
    from svsdct import pipeline
    assert inspect.signature(pipeline.FramePipeline.__init__).parameters["mode"].default is None
    # ... and no module of the package or bench.py spells a mode name next to `or` / as a fallback
    pkg = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
    files = [os.path.join(REPO, "bench.py")] + [os.path.join(r, f) for r, _, fs in os.walk(pkg) for f in fs if f.endswith(".py")]
    for path in files:
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.BoolOp) and isinstance(node.op, ast.Or):
                for v in node.values:
                    assert not (isinstance(v, ast.Constant) and v.value in ("fast", "exact", "guarded")), \
                        f"{path}:{node.lineno}: a mode default outside batch.resolve_mode"
            if isinstance(node, ast.keyword) and node.arg == "mode" and isinstance(node.value, ast.Constant) \
                    and node.value.value in ("fast", "guarded") and not path.endswith("bench.py"):
                raise AssertionError(f"{path}:{node.lineno}: hard-coded mode")
    # the drop-in loops hand the pipeline the resolved default
    for mod in ("embed_process.py", "extract_process.py"):
        text = open(os.path.join(pkg, mod)).read()
        assert "host_level_mode()" in text and '"exact"' not in text
    bench = open(os.path.join(REPO, "bench.py")).read()
    assert "batch.resolve_mode(args.mode)" in bench


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    import subprocess
    exe = str(tmp_path / "use_header")
    libdir = os.path.dirname(native.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "capi_c", "use_header.c"), "-o", exe, "-L" + libdir, "-lsvsdct",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "c abi ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_product_library_reads_no_environment_variable():
    """the shipped library does not even import getenv - experiment knobs (kernel rerouting, occupancy
    caps, SVS_GUARD_SCALE ...) exist only in lib/variants/libsvsdct_exp.so, built with -DSVS_EXPERIMENTS"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in out or "hipModuleLaunchKernel" in out or "__hipPushCallConfiguration" in out   # nm really listed imports
    assert "getenv" not in out
    src = open(os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd", "csrc", "svs_capi.hip")).read()
    outside = re.sub(r"#if defined\(SVS_EXPERIMENTS\).*?#e(?:lse|ndif)", "", src, flags=re.S)
    assert "getenv(" not in outside


def test_capacity_arithmetic_without_gpu():
    lib = native.load()
    p = native.Planes.contiguous(600, 2160, 3840)
    assert lib.svs_capacity_bits(C.byref(p), 3) == 600 * 129600 * 3 == batch.capacity_bits(600, 2160, 3840, 3)
    assert lib.svs_capacity_bits(C.byref(p), 100) == 600 * 129600 * 63       # clamp (config_and_setup.py:138)
    assert lib.svs_capacity_bits(C.byref(p), -4) == 0
    assert lib.svs_packed_bytes(9) == 2


def test_product_path_has_no_cpu_fallback(monkeypatch):
    """Without a GPU the operator must fail loudly, not compute on the CPU."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    import config_and_setup as cs
    with pytest.raises(native.SvsNativeError):
        cs.proses_frame_qim_dct(np.zeros((16, 16), np.uint8), "extract", 8, num_ac_coeffs_to_use=3)
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libsvsdct.so")
    with pytest.raises(native.SvsNativeError, match="no CPU fallback"):
        native.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "hostemu" not in text.replace(
                    "tests/hostemu", ""), os.path.join(root, f)


def test_only_tests_smoke_and_bench_touch_the_oracle():
    """oracle/ is test infrastructure: besides tests/, only __graft_entry__.smoke() and bench.py's baseline /
    checker legs may import it."""
    allowed = {os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")}
    for root, dirs, files in os.walk(REPO):
        dirs[:] = [d for d in dirs if d not in (".git", "tests", "oracle", "gpurun_out", "__pycache__", ".hypothesis",
                                                ".pytest_cache")]
        for f in files:
            path = os.path.join(root, f)
            if f.endswith(".py") and path not in allowed:
                text = open(path).read()
                assert "from oracle" not in text and "import oracle" not in text, path


def test_bit_string_forms():
    import config_and_setup as cs
    assert cs.bytes_ke_bitstream(b"\x80\x01\xff") == "100000000000000111111111"
    assert cs.bitstream_ke_bytes("1000000000000001") == b"\x80\x01"
    assert cs.bitstream_ke_bytes("10000000" + "101") == b"\x80"          # trailing partial byte dropped
    with pytest.raises(ValueError, match="kosong"):
        cs.bitstream_ke_bytes("101")
    assert cs.int_ke_bitstream(5, 8) == "00000101"
    with pytest.raises(ValueError):
        cs.int_ke_bitstream(256, 8)
    assert cs.bitstream_ke_int("00000101", 8) == 5
    with pytest.raises(ValueError):
        cs.bitstream_ke_int("101", 8)
    with pytest.raises(ValueError):
        cs.bitstream_ke_int("")
    bits = synth.synthetic_bits(77)
    assert np.array_equal(batch.str_to_bits(batch.bits_to_str(bits)), bits)
    assert batch.unpack_to_str(batch.pack_bits(bits), 77) == batch.bits_to_str(bits)
    assert batch.pack_bits(bits).size % 4 == 0
    assert batch.str_to_bits("1" * 100, 7).size == 7


def test_frame_sharding_covers_everything_in_order():
    for n, world in [(600, 8), (7, 3), (3, 8), (0, 2)]:
        spans = [batch.shard_frames(n, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (a, ca), (b, _) in zip(spans, spans[1:]):
            assert a + ca == b


def test_operator_rejects_bad_rank_before_touching_the_gpu():
    import config_and_setup as cs
    with pytest.raises(ValueError, match="Format frame input tidak didukung."):
        cs.proses_frame_qim_dct(np.zeros((8, 8, 4), np.uint8), "extract", 8)
