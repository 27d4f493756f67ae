import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd.load()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.mpn_debug_set_gemm_ablate(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib.mpn_debug_set_conv_mode(mode)
if mode == 0: lib.mpn_debug_set_conv_split(1)
for (ci, co, h, w, note) in [(128, 128, 300, 500, "VGG conv2_2: 1200 blocks, padded cols"), (128, 128, 256, 512, "exact tiles, 1024 blocks = 4.0 rounds"),
                             (128, 128, 128, 512, "exact, 512 blocks = 2 rounds"), (128, 128, 64, 512, "exact, 256 blocks = 1 round"),
                             (512, 128, 64, 512, "1 round, 64 chunks"), (512, 128, 256, 512, "4 rounds, 64 chunks"), (64, 128, 256, 512, "4 rounds, 8 chunks")]:
    ms = C.c_float()
    lib.mpn_debug_bench_conv(ci, co, h, w, 0, 10, C.byref(ms))
    print("conv %3d->%3d %3dx%-3d %-40s %8.1f us  %6.1f TF/s" % (ci, co, h, w, note, ms.value * 1e3, 2.0 * h * w * ci * 9 * co / ms.value / 1e9))
