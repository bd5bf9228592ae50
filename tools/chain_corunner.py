"""Diagnostic: one context's chain on a fixed input while something else runs on another stream: does the record change?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amaranth_twstft_amd import _lib as L, frontend, prn
from amaranth_twstft_amd.correlator import Correlator, band_godual
lib = L.load()
dev = torch.device("cuda", 0)
N = 5_000_000; dec = 14; FS = 5e6
taps = frontend.lowpass_taps(70e6, 2.1e6, 0.4e6)
n_in = (N - 1) * dec + taps.size
chips = prn.lfsr_chips(22, 3, 2_500_000)
g = torch.Generator(device=dev); g.manual_seed(1)
cap = (torch.randn((n_in, 2), device=dev, generator=g) * 4000).clamp_(-32768, 32767).to(torch.int16)
win = [(torch.randn((N, 2), device=dev, generator=g) * 4000).to(torch.int16) for _ in range(2)]
big = torch.randn((64 * 1024 * 1024,), device=dev); big2 = torch.empty_like(big)
ma = torch.randn((8192, 8192), device=dev, dtype=torch.float16); mb = torch.randn((8192, 8192), device=dev, dtype=torch.float16)
torch.cuda.synchronize()
band = L.twx_band(*band_godual(FS, N))
key = lambda r: (int(r.indice0), r.xval[0], r.xval[1], r.df)
side = torch.cuda.Stream(device=dev)
sn = 400_000 * 96 + 64
sx = torch.randint(-3000, 3000, (sn, 2), dtype=torch.int16, device=dev)
srep = (torch.randint(0, 2, (400_000,), device=dev).float() * 2 - 1).contiguous()
sout = torch.empty((96, 57, 2), dtype=torch.float64, device=dev)
sink = torch.zeros(4, device=dev)
poison = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'liblds_poison.so'))
poison.lds_poison.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int]
poison.mfma_burn_lds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
poison.mfma_burn_live.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
poison.mfma_burn32.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
poison.mfma_burn.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]
lib.twx_stream.restype = C.c_void_p
with Correlator(chips, fs=FS, Nint=1) as c, Correlator(chips, fs=FS, Nint=1) as b1:
    out = torch.zeros((N, 2), dtype=torch.int16, device=dev)
    def chain(i, r):
        L.check(lib.twx_process_windows_dev(c._h, win[i % 2].data_ptr(), 1, 1, 0, C.byref(band), None, r.data_ptr()), c._h)
    alone = []
    for i in range(2):
        r = torch.zeros(C.sizeof(L.twx_result), dtype=torch.uint8, device=dev); chain(i, r); c.synchronize()
        alone.append(key(L.twx_result.from_buffer_copy(r.cpu().numpy().tobytes())))
    for mode in (os.environ.get("MODES", "nothing,copy,matmul f16,vector FIR,matrix-core FIR,matrix-core FIR x2").split(",")):
        res = torch.zeros((12, C.sizeof(L.twx_result)), dtype=torch.uint8, device=dev)
        os.environ["TWX_FIR_MFMA"] = "0" if mode == "vector FIR" else "1"
        for i in range(12):
            if mode == "copy":
                with torch.cuda.stream(side): big2.copy_(big)
            elif mode == "matmul f16":
                with torch.cuda.stream(side): torch.matmul(ma, mb)
            elif mode.startswith("poison"):
                # the whole LDS of every CU filled with a pattern before the chain (same stream: the chain's kernels start on poisoned LDS)
                pat = {"poison nan": 0x7FC00000, "poison big": 0x7149F2CA, "poison zero": 0}[mode]
                assert poison.lds_poison(C.c_void_p(int(lib.twx_stream(c._h))), pat, 160, 256, 0) == 0
            elif mode.startswith("corun"):
                # LDS-holding workgroups (52 KB each, 2 per CU) of another stream, spinning beside the chain's kernels
                pat = 0x7FC00000 if mode.endswith("nan") else 0x7149F2CA
                assert poison.lds_poison(C.c_void_p(int(lib.twx_stream(b1._h))), pat, 52, 512, 40) == 0
            elif mode.startswith("sliding"):
                # the product's own fp32 matrix-core kernel (k_sliding_mfma: 96 codes, +-28 lags) or its packed-FMA form on another context's stream
                os.environ["TWX_SLIDING_MFMA"] = "1" if mode.endswith("mfma") else "0"
                for _ in range(3):
                    b1.sliding_dot_dev(sx.data_ptr(), sn, srep.data_ptr(), 400_000, 96, 28, sout.data_ptr(), ff=1.234e-5, scale=1.0 / 32768)
            elif mode.startswith("burn lds"):
                src = cap.data_ptr() if mode.endswith("global") else None
                assert poison.mfma_burn_lds(C.c_void_p(int(lib.twx_stream(b1._h))), 512, 1500, src, int(n_in // 4 - 8), sink.data_ptr()) == 0
            elif mode.startswith("burn live"):
                assert poison.mfma_burn_live(C.c_void_p(int(lib.twx_stream(b1._h))), 2048, 3000, sink.data_ptr()) == 0
            elif mode.startswith("burn32"):
                assert poison.mfma_burn32(C.c_void_p(int(lib.twx_stream(b1._h))), 0.0 if mode.endswith("zero") else 1.25, 0.0 if mode.endswith("zero") else 0.75, 2048, 3000, sink.data_ptr()) == 0
            elif mode.startswith("burn"):
                # fp16 matrix-core work on register operands in small workgroups that share CUs with the chain: finite values / zeros / NaN
                ab, bb = {"burn finite": (0x3C003C00, 0x40004000), "burn small": (0x04000400, 0x04000400), "burn zero": (0, 0), "burn nan": (0x7E007E00, 0x7E007E00),
                          "burn random": (0x5A3C2B17, 0xB91E4C63)}[mode]
                assert poison.mfma_burn(C.c_void_p(int(lib.twx_stream(b1._h))), ab, bb, 2048, 6000, sink.data_ptr()) == 0
            elif "FIR" in mode:
                for _ in range(2 if mode.endswith("x2") else 1):
                    b1.fir_decimate_dev(cap.data_ptr(), n_in, taps, dec, out_i16_dev=out.data_ptr())
            chain(i, res[i])
        c.synchronize(); b1.synchronize(); torch.cuda.synchronize()
        bad = [i for i in range(12) if key(L.twx_result.from_buffer_copy(res[i].cpu().numpy().tobytes())) != alone[i % 2]]
        print(mode, "-> wrong records:", bad, flush=True)
