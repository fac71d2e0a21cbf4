"""the re-authored bench scripts keep the reference's helper functions as importable callables with
the reference signatures (kernels/hgemm/tools/utils.py:116-132 try_load_hgemm_library, kernels/hgemm/hgemm.py:84-192
run_benchmark, kernels/flash-attn/flash_attn_mma.py:229-347 run_benchmark, :401-427 check_all_close), and on the GPU
box they run end to end: every row of `--check` must print "all close: True"."""
import inspect
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = os.path.join(ROOT, "cuda-learn-notes_amd", "kernels")


def load(path, name):
    import importlib.util
    for d in (os.path.dirname(path), K):
        if d not in sys.path:
            sys.path.insert(0, d)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def hgemm_script(built):
    return load(os.path.join(K, "hgemm", "hgemm.py"), "cln_hgemm_script")


@pytest.fixture(scope="module")
def fa_script(built):
    return load(os.path.join(K, "flash-attn", "flash_attn_mma.py"), "cln_fa_script")


def params(fn):
    return list(inspect.signature(fn).parameters)


def test_hgemm_script_keeps_the_reference_helpers(hgemm_script):
    h = hgemm_script
    assert params(h.run_benchmark) == ["perf_func", "a", "b", "tag", "out", "stages", "swizzle", "swizzle_stride", "warmup",
                                       "iters", "show_matrix", "only_show_improved"]
    assert params(h.try_load_hgemm_library) == ["force_build", "verbose"]
    for n in ("get_args", "make_block_swizzle_stride", "get_topk_tflops", "get_best_tflops", "plot_tflops", "get_mnk",
              "as_col_major", "get_device_name", "pretty_print_line"):
        assert callable(getattr(h, n)), n
    assert h.make_block_swizzle_stride(4096, 4096) == 2048 and h.make_block_swizzle_stride(256, 256) == 1
    assert h.get_mnk(4096)[0] == [4096, 8192, 12288, 16384]  # range(sep, MMNK + sep, sep), MMNK = 12800


def test_hgemm_run_benchmark_protocol_on_cpu(hgemm_script, capsys):
    """run_benchmark times any callable with the kernel-row signature; returns (out, mean_time_ms) and prints the
    reference row format."""
    h = hgemm_script
    a, b = torch.randn(64, 32).half(), torch.randn(32, 48).half()
    c = torch.zeros(64, 48).half()
    calls = []

    def fake_kernel(x, y, out, stages, swizzle, stride):
        calls.append((stages, swizzle, stride))
        out.copy_((x.float() @ y.float()).half())

    h.MAX_TFLOPS = -1
    out, ms = h.run_benchmark(fake_kernel, a, b, "(mma2x4+warp4x4x2+stage2+dsmem+swizzle<block>)", c, stages=2, swizzle=True,
                              warmup=1, iters=3)
    assert out is c and float(ms) > 0 and len(calls) == 4
    assert calls[0] == (2, False, 1)  # N = 48: stride below 256 switches the block swizzle off (reference :97-99)
    line = capsys.readouterr().out
    assert re.search(r"\(mma2x4\+warp4x4x2\+stage2\+dsmem\+swizzle<block>\): \['.*', '.*'\], time:.*ms, swizzle<block>: NOOP, "
                     r"TFLOPS: .*\(\+0\.00%\)", line)


def test_hgemm_plot_flops_writes_a_chart(hgemm_script, tmp_path):
    h = hgemm_script
    h.args.plot_flops, h.args.save_dir, h.args.save_tag = True, str(tmp_path), "t"
    h.STATIS_INFO.clear(), h.TOATL_TFLOPS.clear()
    h.STATIS_INFO["MNK"] = [256, 512]
    for tag, vals in (("(a)", [1.0, 2.0]), ("(b)", [1.5, 1.0]), ("(cublas)", [2.0, 2.5])):
        h.STATIS_INFO[tag] = vals
        if "cublas" not in tag:
            h.TOATL_TFLOPS[tag] = sum(vals)
    assert h.get_best_tflops() == [1.5, 2.0]
    path = h.plot_tflops()
    assert os.path.getsize(path) > 500 and path.startswith(str(tmp_path))
    from _svgplot import line_chart  # the dependency-free writer used when matplotlib is absent
    svg = line_chart(str(tmp_path / "x.svg"), "t", ["256", "512"], [("(a)", [1.0, 2.0], "dash"), ("(best)", [1.5, 2.0], "bold")])
    assert "<polyline" in open(svg).read()
    h.args.plot_flops = False


def test_flash_attn_script_keeps_the_reference_helpers(fa_script, capsys):
    f = fa_script
    assert params(f.run_benchmark) == ["perf_func", "q", "k", "v", "tag", "out", "s", "stages", "warmup", "iters",
                                       "show_matrix", "only_show_improved"]
    assert params(f.check_all_close) == ["out_flash_or_sdpa", "out_mma", "tag", "check_all", "is_flash"]
    assert params(f.get_qkvo) == ["B", "H", "N", "D"] and params(f.sdpa) == ["q", "k", "v", "use_flash"]
    assert f.MAX_HEADDIM_CFG["mma(split-kv+stage1)"] == 128 and f.MAX_HEADDIM_CFG["mma(split-q+tiling-qkv+stage2)"] == 1024
    assert f.MAX_HEADDIM_CFG["mma(split-q+share-qkv+stage1)"] == 256 and f.MAX_HEADDIM_CFG["mma(split-q+share-qkv+stage2)"] == 128
    x = torch.randn(1, 2, 64, 32).half()
    assert f.check_all_close(x, x + 0.001, "out_mma", False, False) is True
    assert f.check_all_close(x.transpose(1, 2).contiguous(), x + 0.1, "out_mma", False, True) is False
    assert f.check_all_close(None, x) is None
    out = capsys.readouterr().out
    assert "out_sdpa vs out_mma" in out and "all close: True" in out and "out_flash vs out_mma" in out
    # a row above its head-dim limit, or filtered by a flag, is skipped exactly like the reference does
    q = torch.zeros(1, 1, 64, 256).half()
    assert f.run_benchmark(lambda *a: None, q, q, q, "mma(split-kv+stage1)", q.clone(), stages=1) == (None, None)
    assert f.run_benchmark(lambda *a: None, q, q, q, "(sdpa)") == (None, None)  # --sdpa not given


def run_script(rel, *argv):
    env = dict(os.environ, CLN_AMD_SEED="0")
    r = subprocess.run([sys.executable, os.path.join(K, rel)] + list(argv), capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_hgemm_script_runs_on_the_gpu(built, dev):
    out = run_script("hgemm/hgemm.py", "--mma", "--MNK", "1024", "--show-all-info")
    rows = re.findall(r"^\s+(\S*\(.*\)): \['(.*)', '(.*)'\], time:.*TFLOPS: ([\d.]+)", out, flags=re.M)
    assert len(rows) >= 15, out[-1500:]
    vals = {(r[1], r[2]) for r in rows if "cublas" not in r[0]}
    cub = [(r[1], r[2]) for r in rows if r[0] == "(cublas)"]
    assert cub, "vendor row missing"
    assert any(r[0] == "(hipblaslt)" for r in rows), "hipBLASLt row missing"
    # every kernel row prints the same first/last element of C as the vendor row, to fp16 rounding of |C| ~ 32
    for (a, b) in vals:
        assert abs(float(a) - float(cub[0][0])) <= 0.13 and abs(float(b) - float(cub[0][1])) <= 0.13, (a, b, cub[0])
    assert all(float(r[3]) > 1.0 for r in rows)


@pytest.mark.gpu
def test_flash_attn_script_check_all_rows_close(built, dev):
    out = run_script("flash-attn/flash_attn_mma.py", "--B", "1", "--H", "8", "--N", "1024", "--D", "64", "--check",
                     "--show-all", "--seed", "1", "--iters", "2")
    verdicts = re.findall(r"out_sdpa vs (\S+)\s*, all close: (\w+)", out)
    assert len(verdicts) >= 30, out[-2000:]  # 2 stages x (split-kv, split-q, share-kv x4, share-qkv x4, tiling x 8)
    assert all(v == "True" for _, v in verdicts), [t for t, v in verdicts if v != "True"]
    assert any("split-kv" in t for t, _ in verdicts)


@pytest.mark.gpu
def test_flash_attn_script_config_c5_rows(built, dev):
    """D = 512 (C5 head dim, shorter sequence): only the tiling rows run (head-dim table), all close to SDPA."""
    out = run_script("flash-attn/flash_attn_mma.py", "--B", "1", "--H", "4", "--N", "512", "--D", "512", "--check",
                     "--show-all", "--seed", "2", "--iters", "1")
    verdicts = re.findall(r"out_sdpa vs (\S+)\s*, all close: (\w+)", out)
    assert verdicts and all("tiling" in t for t, _ in verdicts)
    assert all(v == "True" for _, v in verdicts), verdicts
