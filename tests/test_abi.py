"""The C-ABI library builds for gfx950, loads on a CPU-only host and exports every symbol include/densepose_hip.h declares.
(No compute calls here: those need a GPU and live in the -m gpu tests.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from densepose_torchscript_amd import lib
    lib.build_library()
    return lib


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "densepose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(built):
    declared = _declared_symbols()
    assert declared == sorted(built.SYMBOLS), (declared, sorted(built.SYMBOLS))


def test_library_exports_every_declared_symbol(built):
    raw = ctypes.CDLL(built.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), name
    lib = built.load()
    assert lib.dp_abi_version() == 8 == built.ABI_VERSION


def test_argument_validation_without_gpu(built):
    """Bad arguments are rejected on the host before anything is launched (error code + message)."""
    lib = built.load()
    p = built.ConvParams()
    p.N, p.H, p.W, p.Ho, p.Wo, p.Cin, p.Cout = 1, 4, 4, 4, 4, 7, 8
    rc = lib.dp_conv2d_nhwc(ctypes.byref(p), None)
    assert rc == -1 and b"null pointer" in lib.dp_last_error()
    b = built.BottleneckParams()
    b.N, b.H, b.W, b.Cmid, b.Cout, b.Kpad2, b.Kpad3, b.ntaps2, b.k_order2, b.dtype = 1, 8, 8, 64, 256, 576, 64, 9, 1, built.DP_F32
    assert lib.dp_bottleneck_tail_supported(ctypes.byref(b)) == 0          # fp32 parity mode has no fused kernel
    assert lib.dp_bottleneck_tail_nhwc(ctypes.byref(b), None) == -2 and b"16-bit" in lib.dp_last_error()
    b.dtype = built.DP_BF16
    assert lib.dp_bottleneck_tail_supported(ctypes.byref(b)) == 1
    assert lib.dp_bottleneck_tail_nhwc(ctypes.byref(b), None) == -1 and b"null pointer" in lib.dp_last_error()
    b.N = 64                                                                 # 64 x 200 x 336 pixels x 512 B > 2^31: caller must chunk
    b.H, b.W = 200, 336
    assert lib.dp_bottleneck_tail_supported(ctypes.byref(b)) == 0
    q = built.NmsParams()
    assert lib.dp_batched_nms(ctypes.byref(q), None) == -1
    assert lib.dp_nms_workspace_bytes(2, 1000) > 2 * 1000 * 16 * 8
    assert lib.dp_rpn_topk_workspace_bytes(8, 200, 336, 3) >= 8 * 200 * 336 * 3 * 4


def test_struct_layout_matches_c(built, tmp_path):
    """sizeof() of every parameter struct as seen by a C compiler == ctypes.sizeof of its mirror."""
    import subprocess
    names = {"dp_preprocess_params": built.PreprocessParams, "dp_conv_params": built.ConvParams,
             "dp_bottleneck_params": built.BottleneckParams, "dp_stem_pool_params": built.StemPoolParams,
             "dp_pack_params": built.PackParams, "dp_pack_info": built.PackInfo,
             "dp_rpn_level_params": built.RpnLevelParams, "dp_nms_params": built.NmsParams,
             "dp_roi_align_params": built.RoiAlignParams, "dp_box_decode_params": built.BoxDecodeParams,
             "dp_postprocess_params": built.PostprocessParams, "dp_iuv_params": built.IuvParams,
             "dp_groupnorm_params": built.GroupNormParams, "dp_resize_params": built.ResizeParams,
             "dp_iuv_extract_params": built.IuvExtractParams, "dp_pair_params": built.PairParams}
    src = '#include <stdio.h>\n#include "densepose_hip.h"\nint main(){' + "".join(
        'printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}"
    c = tmp_path / "s.c"
    c.write_text(src)
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n, cls in names.items():
        assert sizes[n] == ctypes.sizeof(cls), (n, sizes[n], ctypes.sizeof(cls))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "densepose_torchscript_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M), f
                assert "/root/reference" not in re.sub(r'""".*?"""', "", s, flags=re.S).replace("# ", ""), f


def test_oracle_restates_the_network_structure_itself():
    """The converse for the two structure helpers: the checker (oracle/ref_cpu.py, oracle/ref_storage.py) walks the block table and the
    decoder layout of oracle/structure.py (restated from resnet.py:641-688 / roi_head.py:42-68), never the product's
    densepose_torchscript_amd.weights.resnet_blocks / decoder_layout - and the two restatements agree on every BASELINE configuration."""
    from densepose_torchscript_amd import get_config
    from densepose_torchscript_amd import weights as W
    from oracle import structure as S
    for f in ("ref_cpu.py", "ref_storage.py", "ops_ref.py", "structure.py"):
        src = open(os.path.join(ROOT, "oracle", f)).read()
        assert not re.search(r"densepose_torchscript_amd\.weights\s+import\s+[^\n]*(resnet_blocks|decoder_layout)", src), f
    for name in ("densepose_rcnn_R_50_FPN_s1x_legacy", "densepose_rcnn_R_50_FPN_s1x", "densepose_rcnn_R_101_FPN_s1x",
                 "densepose_rcnn_R_50_FPN_DL_s1x", "densepose_rcnn_R_101_FPN_DL_s1x"):
        cfg = get_config(name, [])
        assert list(S.resnet_blocks(cfg)) == list(W.resnet_blocks(cfg)), name
        assert list(S.decoder_layout(cfg)) == list(W.decoder_layout(cfg)), name
    r101 = S.resnet_blocks(get_config("densepose_rcnn_R_101_FPN_s1x", []))
    assert [sum(1 for b in r101 if b[0] == "res%d" % k) for k in (2, 3, 4, 5)] == [3, 4, 23, 3]       # resnet.py:641-647
    assert [(b[0], b[5], b[6]) for b in r101 if b[1] == 0] == [("res2", 1, True), ("res3", 2, True), ("res4", 2, True), ("res5", 2, True)]


def test_no_kernel_uses_scratch(built, tmp_path):
    """No gfx950 kernel of the library spills or keeps a private array: zero scratch_ instructions in the disassembly and a
    zero .private_segment_fixed_size in every kernel descriptor (round 3 shipped the two-source LDS-ring instances with 20 bytes
    of scratch and scratch loads between the MFMAs of the steady loop)."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = os.path.join(llvm, "llvm-objdump"), os.path.join(llvm, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm llvm tools not installed")
    so = tmp_path / "lib.so"
    shutil.copy(built.LIB_PATH, so)        # --offloading extracts the code objects beside its input
    subprocess.check_call([objdump, "--offloading", str(so)], stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    objs = sorted(p for p in tmp_path.iterdir() if "amdgcn" in p.name)
    assert objs, "no gfx950 code object found in %s" % built.LIB_PATH
    n_kernels = 0
    for o in objs:
        dis = subprocess.check_output([objdump, "-d", str(o)]).decode()
        bad = [ln for ln in dis.splitlines() if "scratch_" in ln]
        assert not bad, (o.name, len(bad), bad[:3])
        notes = subprocess.check_output([readelf, "--notes", str(o)]).decode()
        sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes)
        names = re.findall(r"\.name:\s*(\S+)", notes)
        n_kernels += len(sizes)
        assert sizes and all(int(s) == 0 for s in sizes), (o.name, [(n, s) for n, s in zip(names, sizes) if int(s)])
    assert n_kernels > 100     # every template instance of every kernel was looked at


def test_policy_table_and_no_environment_reads(built, monkeypatch):
    """Kernel choice is a property of the layer and of the policy table (dp_set_policy), never of the process environment: the
    library's sources contain no getenv, an environment variable set AFTER the load changes nothing, the table does."""
    import glob
    import os
    for f in glob.glob(os.path.join(built.CSRC, "*")):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(f).read(), f
    lib = built.load()
    built.reset_policy()
    keys = built.policy_keys()
    assert "conv_pws" in keys and "conv_big" in keys and len(keys) == len(set(keys))
    assert lib.dp_set_policy(b"no_such_key", 1) == -1
    p = built.ConvParams()       # res4's conv1 (resnet.py:192-193): 1024 -> 256 pointwise on 8 x 50 x 84 pixels, bf16
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = 8, 50, 84, 1024, 50, 84, 256, 256, 1024
    p.stride, p.ntaps, p.dtype = 1, 1, built.DP_BF16
    p.osN, p.osH, p.osW = 50 * 84 * 256, 84 * 256, 256
    p.in_, p.weight, p.out = 4096, 4096, 4096      # aligned placeholders: the class query looks at NULL / alignment only
    assert lib.dp_conv2d_kernel_class(ctypes.byref(p)) == 9
    p.N = 1                                        # the class of a layer does not depend on the batch
    assert lib.dp_conv2d_kernel_class(ctypes.byref(p)) == 9
    monkeypatch.setenv("DP_CONV_PWS", "0")         # the environment is not consulted after the load ...
    assert lib.dp_conv2d_kernel_class(ctypes.byref(p)) == 9
    with built.policy(conv_pws=0):                 # ... the table is
        assert built.get_policy("conv_pws") == 0
        assert lib.dp_conv2d_kernel_class(ctypes.byref(p)) in (2, 3)
    assert lib.dp_conv2d_kernel_class(ctypes.byref(p)) == 9
    p.dtype = built.DP_F32                         # fp32 parity mode stays on the exact-fp32 kernels
    assert lib.dp_conv2d_kernel_class(ctypes.byref(p)) != 9


def test_product_reads_no_environment_for_its_switches(monkeypatch):
    """The A/B switches of the host side are constructor arguments (options.EngineOptions), not environment variables: the only os.environ uses
    under densepose_torchscript_amd/ are the policy-table variables lib.py applies ONCE at load for the command-line tools, the rank variables
    of torch.distributed in parallel.py, and EngineOptions.from_env - which the tools call explicitly."""
    import re
    pkg = os.path.join(ROOT, "densepose_torchscript_amd")
    allowed = {"lib.py", "parallel.py", "options.py"}
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py") and fn not in allowed:
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"os\.environ|getenv", src), fn
    from densepose_torchscript_amd.options import EngineOptions
    assert EngineOptions() == EngineOptions.from_env({})
    monkeypatch.setenv("DP_FUSE_PAIR", "0")
    assert EngineOptions().fuse_pair is True                      # the default does not look at the environment ...
    o = EngineOptions.from_env()
    assert o.fuse_pair is False and o.fork_levels == 0           # ... the tools' constructor does
    assert EngineOptions.from_env({"DP_FORK": "3", "DP_DECODER_FOLD": "0"}) == EngineOptions(fork_levels=3, decoder_fold=False)


def test_entry_scripts_have_no_undefined_names():
    """bench.py / run.py / __graft_entry__.py only run on the GPU box: a name used in a function but imported in another one (a round-6 slip in
    bench.py: EngineOptions) would show up there first, as a crashed benchmark. A scope-aware static check: every name a function loads is bound in
    that function, at module level, or a builtin."""
    import ast
    import builtins
    for fn_ in ("bench.py", "run.py", "__graft_entry__.py"):
        tree = ast.parse(open(os.path.join(ROOT, fn_)).read())
        mod = set(dir(builtins)) | {"__file__", "__name__"}
        for n in tree.body:
            if isinstance(n, (ast.Import, ast.ImportFrom)):
                mod |= {(a.asname or a.name).split(".")[0] for a in n.names}
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                mod.add(n.name)
            elif isinstance(n, (ast.Assign, ast.AugAssign, ast.AnnAssign, ast.If, ast.Try, ast.With, ast.For)):
                mod |= {x.id for x in ast.walk(n) if isinstance(x, ast.Name) and isinstance(x.ctx, ast.Store)}
                mod |= {(a.asname or a.name).split(".")[0] for x in ast.walk(n) if isinstance(x, (ast.Import, ast.ImportFrom)) for a in x.names}
        for f in (n for n in tree.body if isinstance(n, ast.FunctionDef)):
            bound = set(mod)
            for n in ast.walk(f):
                if isinstance(n, (ast.Import, ast.ImportFrom)):
                    bound |= {(a.asname or a.name).split(".")[0] for a in n.names}
                elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                    bound.add(n.name)
                elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                    bound.add(n.id)
                elif isinstance(n, ast.arg):
                    bound.add(n.arg)
                elif isinstance(n, ast.ExceptHandler) and n.name:
                    bound.add(n.name)
            used = {n.id for n in ast.walk(f) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
            assert not (used - bound), (fn_, f.name, sorted(used - bound))
