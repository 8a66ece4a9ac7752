// Probe: where does conv_igemm_kernel<3> spend its time?  The product kernel is compiled here with parts removed
// (DFINE_CONV3X3_ABLATE bits: 1 no MFMAs, 2 no LDS fragment reads, 4 no staging of x into LDS, 8 no weight loads) and timed
// on the layer shapes given on the command line (Cin Cout side); results are garbage in the ablated builds.
//   for a in 0 1 2 4 8 ...; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDFINE_CONV3X3_ABLATE=$a -Icustom_d_fine_amd/csrc -Iinclude \
//       -o tools/probe/_bin/c3abl$a tools/probe/conv3x3_ablate.hip; tools/probe/_bin/c3abl$a 128 128 80  128 128 40  32 32 160; done
#include "../../custom_d_fine_amd/csrc/conv.hip"
#include <cstdio>
#include <cstdlib>

namespace dfine { void set_last_error(hipError_t) {} }   // defined in msda.hip of the library

int main(int argc, char **argv) {
    const int B = 32;
    for (int i = 1; i + 2 < argc; i += 3) {
        const int Cin = atoi(argv[i]), Cout = atoi(argv[i + 1]), side = atoi(argv[i + 2]), HW = side * side;
        const int NP = (Cout + 15) / 16 * 16, KP = (Cin + 31) / 32 * 32;
        uint16_t *x, *w, *y;
        hipMalloc(&x, (size_t)B * Cin * HW * 2 + (1 << 20));
        hipMalloc(&w, (size_t)9 * (NP + 128) * KP * 2);
        hipMalloc(&y, (size_t)B * Cout * HW * 2 + (1 << 20));
        hipMemset(x, 0x11, (size_t)B * Cin * HW * 2);
        hipMemset(w, 0x22, (size_t)9 * (NP + 128) * KP * 2);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int k = 0; k < 3; ++k) dfine::launch_conv(x, w, y, B, Cin, Cout, NP, KP, side, side, 3, 0);
        hipDeviceSynchronize();
        const int reps = 20;
        hipEventRecord(e0, 0);
        for (int k = 0; k < reps; ++k) dfine::launch_conv(x, w, y, B, Cin, Cout, NP, KP, side, side, 3, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, fl = 18.0 * B * HW * Cin * Cout, by = 2.0 * B * HW * (Cin + Cout);
        printf("ablate %d  %4d -> %4d @ %5d px: %7.1f us  (%6.0f TFLOP/s, %5.2f TB/s of x + y)\n", DFINE_CONV3X3_ABLATE, Cin, Cout, HW, us,
               fl / us * 1e-6, by / us * 1e-6);
        hipFree(x); hipFree(w); hipFree(y);
    }
    return 0;
}
