"""CPU: the C-ABI library loads and exports every symbol include/pnp_hip.h declares (no compute
calls without a GPU), and the host-side argument validation that needs no device works."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pnp-ovss_amd", "pnp_ovss", "libpnp_hip.so")


def _declared():
    src = open(os.path.join(ROOT, "include", "pnp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnp_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "pnp-ovss_amd", "csrc"), "-j8"])
    return ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pnp_hip.h but not exported"


def test_binding_table_matches_header():
    from pnp_ovss import hip
    assert sorted(hip.EXPORTED) == _declared()


def test_workspace_bytes_is_host_only(lib):
    from pnp_ovss import hip, config as C
    cfg = C.blip_itm_large(336)
    c = hip.PnpConfig(cfg.img_size, cfg.patch, cfg.vit_dim, cfg.vit_depth, cfg.vit_heads, cfg.vit_mlp_ratio,
                      cfg.vit_ln_eps, cfg.txt_hidden, cfg.txt_layers, cfg.txt_heads, cfg.txt_inter, cfg.txt_ln_eps,
                      cfg.vocab, cfg.max_pos, cfg.enc_token_id, 35, 32, 7, 1, 0)
    lib.pnp_workspace_bytes.restype = ctypes.c_size_t
    n = lib.pnp_workspace_bytes(ctypes.byref(c))
    assert 1 << 30 < n < 64 << 30          # a few GB for batch 35 at 336^2; far below 288 GB HBM


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pnp_ovss import hip, config as C
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.Engine(C.blip_itm_small(64), max_batch=1)


def test_null_engine_calls_return_errors(lib):
    lib.pnp_vit_forward.restype = ctypes.c_int32
    assert lib.pnp_vit_forward(None, None, None, 1, None) != 0
    lib.pnp_last_error.restype = ctypes.c_char_p
    assert lib.pnp_last_error(None) == b"null engine"


def test_operator_entry_points_validate_arguments_without_a_gpu(lib):
    """Argument checks that sit in front of any HIP call: null operands and out-of-range modes are refused (PNP_ERR_ARG = -22)
    on a box without a GPU too."""
    lib.pnp_op_gemm_x3a.restype = ctypes.c_int
    lib.pnp_op_gemm_x3.restype = ctypes.c_int
    n = None
    assert lib.pnp_op_gemm_x3a(n, 64, n, n, 64, 8, 64, 64, n, n, 0, n, 64, 0, n, 0, n) == -22
    assert lib.pnp_op_gemm_x3(n, n, 64, n, n, 64, 8, 64, 64, n, 0, n, 0, n, 0, n, n, 0, 0, 0, 0, n) == -22


def _gfx950_code_objects(lib_path, tmpdir):
    """The gfx950 code objects inside the library's .hip_fatbin section (one clang offload bundle per translation unit:
    magic, u64 entry count, per entry u64 offset / u64 size / u64 triple length / triple)."""
    import struct
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    fat = os.path.join(tmpdir, "fat.bin")
    subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
    d = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], d.find(magic)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", d, at + len(magic))
        p = at + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            triple = d[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(d[at + off: at + off + size])
        at = d.find(magic, at + 1)
    return out


def test_m0_is_written_only_by_the_asm_lds_dma(lib, tmp_path):
    """The split-bf16 GEMM and attention kernels issue their LDS-DMA as inline asm that writes m0 (the LDS base of a piece).
    hipcc reserves m0 and rejects it in a clobber list, so the asm relies on the compiler having no m0 use of its own in those
    kernels.  Checked on the shipped code object: in every gemm_nt_x3 / vit_attn32_x3 kernel each instruction
    that names m0 is `s_mov_b32 m0, <sgpr>` and the next vector-memory instruction behind it is a global_load_lds."""
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    checked = 0
    for i, co in enumerate(_gfx950_code_objects(LIB, str(tmp_path))):
        f = tmp_path / f"co{i}.o"
        f.write_bytes(co)
        asm = subprocess.run([objdump, "-d", str(f)], capture_output=True, text=True, check=True).stdout
        name, body = None, {}
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                body[name] = []
            elif name and line.startswith("\t"):
                body[name].append(line.split("//")[0].strip())
        for k, ins in body.items():
            if "gemm_nt_x3_kernel" not in k and "vit_attn32_x3_kernel" not in k:
                continue
            uses = [j for j, t in enumerate(ins) if re.search(r"\bm0\b", t)]
            assert uses, f"{k}: no m0 write found (the asm LDS-DMA should be there)"
            for j in uses:
                assert re.match(r"s_mov_b32 m0, s\d+$", ins[j]), f"{k}: foreign m0 use `{ins[j]}`"
                nxt = next((t for t in ins[j + 1: j + 6] if t.startswith(("global_", "buffer_", "flat_", "ds_"))), "")
                assert nxt.startswith("global_load_lds_dwordx4"), f"{k}: `{ins[j]}` is followed by `{nxt}`"
            checked += 1
    assert checked >= 6, checked            # five GEMM epilogue variants + the attention kernel(s)
