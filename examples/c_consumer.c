/* Plain-C consumer of libmsiren.so: the drop-in boundary used without Python or torch.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_consumer.c -Lmri_inr_amd -lmsiren -Wl,-rpath,$PWD/mri_inr_amd -lm -o c_consumer
 *   ./c_consumer weights.bin mods.bin out.bin
 *
 * weights.bin: a sequence of records {int32 name_len; char name[name_len]; int64 n; float data[n]} holding
 * the state_dict (reference key names); mods.bin: {int32 L; int32 B; int32 H; float mods[L*B*H]}.
 * Writes out.bin: {int32 B; int32 P; float out[B*P]} = SirenNet.forward on the coordinate grid
 * (src/networks/modulated_siren.py:215-233) through msiren_forward_mods.
 * Exit code 0 = ok, 2 = no usable device (msiren_create refused), 1 = anything else.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msiren.h"

static int die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, msiren_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s weights.bin mods.bin out.bin\n", argv[0]);
        return 1;
    }
    struct msiren_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = MSIREN_ABI_VERSION;
    cfg.dim_in = 2;
    cfg.dim_hidden = 256;
    cfg.dim_out = 1;
    cfg.num_layers = 5;
    cfg.latent_dim = 256;
    cfg.w0 = 1.0f;
    cfg.w0_initial = 30.0f;
    cfg.use_bias = 1;
    cfg.activation = MSIREN_ACT_SINE;
    cfg.outer_patch_size = 32;
    cfg.inner_patch_size = 16;
    cfg.siren_patch_size = 24;
    cfg.precision = MSIREN_PREC_F16X3;
    cfg.device = 0;

    msiren_handle h = NULL;
    if (msiren_create(&cfg, &h) != MSIREN_OK) {
        fprintf(stderr, "msiren_create: %s\n", msiren_last_error());
        return 2;
    }

    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    for (;;) {
        int32_t len;
        char name[256];
        int64_t n;
        if (fread(&len, 4, 1, f) != 1) break;
        if (len <= 0 || len >= (int32_t)sizeof name || fread(name, 1, (size_t)len, f) != (size_t)len) return 1;
        name[len] = 0;
        if (fread(&n, 8, 1, f) != 1 || n < 0) return 1;
        float* data = (float*)malloc((size_t)n * sizeof(float) + 1);
        if (!data || fread(data, sizeof(float), (size_t)n, f) != (size_t)n) return 1;
        if (msiren_set_tensor(h, name, data, (size_t)n) != MSIREN_OK) return die(name);
        free(data);
    }
    fclose(f);
    if (msiren_commit_weights(h) != MSIREN_OK) return die("msiren_commit_weights");

    int32_t dims[3];
    f = fopen(argv[2], "rb");
    if (!f || fread(dims, 4, 3, f) != 3 || dims[0] != cfg.num_layers || dims[2] != cfg.dim_hidden) return 1;
    const int64_t B = dims[1];
    const int32_t P = cfg.siren_patch_size * cfg.siren_patch_size;
    const size_t nm = (size_t)dims[0] * (size_t)B * (size_t)dims[2];
    float* mods = (float*)malloc(nm * sizeof(float));
    float* out = (float*)malloc((size_t)B * (size_t)P * sizeof(float));
    if (!mods || !out || fread(mods, sizeof(float), nm, f) != nm) return 1;
    fclose(f);

    if (msiren_forward_mods(h, mods, B, out) != MSIREN_OK) return die("msiren_forward_mods");

    f = fopen(argv[3], "wb");
    if (!f) return 1;
    const int32_t hdr[2] = {(int32_t)B, P};
    fwrite(hdr, 4, 2, f);
    fwrite(out, sizeof(float), (size_t)B * (size_t)P, f);
    fclose(f);
    double flops = 0.0;
    msiren_flops_per_coord(h, &flops);
    printf("ok: %lld patches x %d coordinates, %.0f FLOP per coordinate\n", (long long)B, P, flops);
    /* which HIP runtime the library's calls ran on: a C host has no torch in the process, so the system one */
    int32_t rt = 0, built = 0;
    char path[512];
    msiren_runtime_info(&rt, &built, NULL, path, sizeof path);
    printf("hip runtime %d (library built against %d): %s\n", rt, built, path);
    msiren_destroy(h);
    free(mods);
    free(out);
    return 0;
}
