"""Kernel-only timing of the per-ROI bf16 convolutions of BASELINE configs[3] / [4] (one tower), layer by layer, through
mpn_debug_bench_conv_bf16 (libmpn_hip_dbg.so): ms, TFLOP/s and fraction of the 2.5 PFLOP/s dense bf16 peak per layer class.
Usage: python tools/bench_conv_bf16.py [inception|resnet|all] [hook=value ...]   e.g.  bf16_exp=1 bf16_dma_tn=256"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")
which = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "all"
CLK = "--clk" in sys.argv   # sample the GPU's sclk / power (bench.ClockSampler) while each layer runs for ~0.4 s
if CLK:
    sys.argv.remove("--clk")
    import torch, bench
    torch.cuda.init()
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        getattr(lib, "mpn_debug_set_" + k)(int(v))
# (name, Cin, Cout, KH, KW, sh, sw, ph, pw, B, H, W, residual, count per tower)
INC = [("7a 1x1 768->384 (fused siblings) @17", 768, 384, 1, 1, 1, 1, 0, 0, 2000, 17, 17, 0, 1),
       ("7a 3x3/2 192->320 @17->8", 192, 320, 3, 3, 2, 2, 0, 0, 2000, 17, 17, 0, 1),
       ("7a 1x7 192->192 @17", 192, 192, 1, 7, 1, 1, 0, 3, 2000, 17, 17, 0, 1),
       ("7a 7x1 192->192 @17", 192, 192, 7, 1, 1, 1, 3, 0, 2000, 17, 17, 0, 1),
       ("7a 3x3/2 192->192 @17->8", 192, 192, 3, 3, 2, 2, 0, 0, 2000, 17, 17, 0, 1),
       ("7b 1x1 1280->1344 (fused) @8", 1280, 1344, 1, 1, 1, 1, 0, 0, 2000, 8, 8, 0, 1),
       ("7b/c 1x3 384->384 @8", 384, 384, 1, 3, 1, 1, 0, 1, 2000, 8, 8, 0, 4),
       ("7b/c 3x1 384->384 @8", 384, 384, 3, 1, 1, 1, 1, 0, 2000, 8, 8, 0, 4),
       ("7b/c 3x3 448->384 @8", 448, 384, 3, 3, 1, 1, 1, 1, 2000, 8, 8, 0, 2),
       ("7c 1x1 2048->1344 (fused) @8", 2048, 1344, 1, 1, 1, 1, 0, 0, 2000, 8, 8, 0, 1)]
RES = [("l4b1 1x1 1024->512 @14", 1024, 512, 1, 1, 1, 1, 0, 0, 1000, 14, 14, 0, 1),
       ("l4b1 3x3/2 512->512 @14->7", 512, 512, 3, 3, 2, 2, 1, 1, 1000, 14, 14, 0, 1),
       ("l4b1 sc 1x1/2 1024->2048 @14->7", 1024, 2048, 1, 1, 2, 2, 0, 0, 1000, 14, 14, 0, 1),
       ("l4 1x1 512->2048 + res @7", 512, 2048, 1, 1, 1, 1, 0, 0, 1000, 7, 7, 1, 3),
       ("l4 1x1 2048->512 @7", 2048, 512, 1, 1, 1, 1, 0, 0, 1000, 7, 7, 0, 2),
       ("l4 3x3 512->512 @7", 512, 512, 3, 3, 1, 1, 1, 1, 1000, 7, 7, 0, 2)]
for name, layers in (("inception", INC), ("resnet", RES)):
    if which not in ("all", name):
        continue
    tot_ms = tot_f = 0.0
    for (nm, ci, co, kh, kw, sh, sw, ph, pw, B, H, W, res, cnt) in layers:
        ms = C.c_float()
        ck = C.c_ulonglong()
        rc = lib.mpn_debug_bench_conv_bf16(ci, co, kh, kw, sh, sw, ph, pw, B, H, W, res, 10, C.byref(ms), C.byref(ck))
        assert rc == 0, (nm, rc)
        clk = ""
        if CLK:
            smp = bench.ClockSampler(0).start()
            rc = lib.mpn_debug_bench_conv_bf16(ci, co, kh, kw, sh, sw, ph, pw, B, H, W, res, max(10, int(400.0 / ms.value)), C.byref(ms), None)
            smp.stop()
            sm = smp.summary()
            clk = "  sclk %s MHz  power %s W" % ((sm["sclk_mhz"] or {}).get("mean"), (sm["power_w"] or {}).get("mean"))
        oh, ow = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        fl = 2.0 * B * oh * ow * ci * kh * kw * co
        print("%-40s %8.1f us  %7.1f TFLOP/s  %.3f of 2.5 PF   (x%d per tower)  ck %016x%s" % (nm, ms.value * 1e3, fl / ms.value / 1e9, fl / ms.value / 1e9 / 2500.0, cnt, ck.value, clk))
        tot_ms += ms.value * cnt; tot_f += fl * cnt
    print("%s tower convolutions: %.3f ms, %.2f TFLOP -> %.1f TFLOP/s = %.3f of 2.5 PF" % (name, tot_ms, tot_f / 1e12, tot_f / tot_ms / 1e9, tot_f / tot_ms / 1e9 / 2500.0))
