"""kernel-only timing of every VGG-16 conv layer (600x1000 input) and the fc layers, in the pipeline's layouts"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.mpn_debug_set_conv_variant(variant); lib.mpn_debug_set_conv_split(split)
cfg = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512, 512]
h, w, cin = 600, 1000, 3
layers = []
for i, it in enumerate(cfg):
    if it == "P": continue
    pool = 1 if (i + 1 < len(cfg) and cfg[i + 1] == "P") else 0
    layers.append((cin, it, h, w, pool)); cin = it
    if pool: h, w = (h + 1) // 2, (w + 1) // 2
tot_ms = tot_fl = 0
for (ci, co, hh, ww, pool) in layers:
    ms = C.c_float()
    rc = lib.mpn_debug_bench_conv(ci, co, hh, ww, pool, 10, C.byref(ms))
    fl = 2.0 * hh * ww * ci * 9 * co
    tot_ms += ms.value; tot_fl += fl
    print("conv %3d->%3d %4dx%-4d pool=%d  %8.1f us  %6.1f TF/s (%4.1f%%) rc=%d" % (ci, co, hh, ww, pool, ms.value * 1e3, fl / ms.value / 1e9, fl / ms.value / 1e9 / 1.573, rc))
print("trunk total %.3f ms  %.1f TF/s" % (tot_ms, tot_fl / tot_ms / 1e9))
for (M, K, N) in [(1000, 25088, 4096), (1000, 4096, 4096), (1000, 4096, 105)]:
    ms = C.c_float()
    rc = lib.mpn_debug_bench_linear(M, K, N, 10, C.byref(ms))
    fl = 2.0 * M * K * N
    print("linear M=%d K=%d N=%d  %8.1f us  %6.1f TF/s (%4.1f%%) rc=%d" % (M, K, N, ms.value * 1e3, fl / ms.value / 1e9, fl / ms.value / 1e9 / 1.573, rc))
