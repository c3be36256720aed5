/* A compiled-C consumer of include/s3r.h — what a maintainer binding libs3r_hip.so from C (or cgo / JNI / N-API, which
 * all go through exactly this header) gets.  Built by tests/test_c_abi_cpu.py with plain `gcc -I include`, linked against
 * the in-tree libs3r_hip.so.  It never touches a GPU: only the planning entries are called.
 *
 * The reference's one native binding is extensions/chamfer_dist (/root/reference/README.md:64-65); this boundary stands
 * in for it and for the nn.Module forwards (README.md:91).
 *
 *   stdin:   desc  <22 fields of s3r_conv_desc in declaration order; act_param as a float>
 *            chain <n>   followed by n `desc` lines
 *            linear <batch> <cin> <cout>
 *   stdout:  layout     one line with sizeof / offsetof of every struct the header declares
 *            desc   ->  "desc <out_size> <packed_rc> <packed_elems> <scratch_elems> <wino_layout> <wino_elems>"
 *            chain  ->  "chain <workspace_elems>"
 *            linear ->  "linear <scratch_elems>"
 */
#include <assert.h>
#include <inttypes.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "s3r.h"

/* ABI 8 layout: a caller that strides s3r_layer[] by any other size corrupts every layer after the first */
_Static_assert(S3R_ABI_VERSION == 8, "this consumer is written against ABI 8");
_Static_assert(sizeof(s3r_conv_desc) == 88, "s3r_conv_desc is 22 four-byte fields");
_Static_assert(sizeof(s3r_layer) == 112, "s3r_layer = desc + three pointers");
_Static_assert(sizeof(s3r_prof_record) == 48, "s3r_prof_record");
#define OFF(f, o) _Static_assert(offsetof(s3r_conv_desc, f) == (o), "offset of s3r_conv_desc." #f)
OFF(op, 0); OFF(ndim, 4); OFF(batch, 8); OFF(cin, 12); OFF(cout, 16); OFF(in_size, 20); OFF(k, 24); OFF(stride, 28);
OFF(pad, 32); OFF(act, 36); OFF(tag, 40); OFF(tile, 44); OFF(in_halo, 48); OFF(out_halo, 52); OFF(ksplit, 56);
OFF(dtype, 60); OFF(in_layout, 64); OFF(out_layout, 68); OFF(algo, 72); OFF(dilation, 76); OFF(out_pad, 80);
OFF(act_param, 84);
_Static_assert(offsetof(s3r_layer, desc) == 0 && offsetof(s3r_layer, packed_w) == 88 && offsetof(s3r_layer, scale) == 96 &&
               offsetof(s3r_layer, shift) == 104, "s3r_layer field offsets");
_Static_assert(offsetof(s3r_prof_record, family) == 0 && offsetof(s3r_prof_record, tag) == 4 &&
               offsetof(s3r_prof_record, ms) == 8 && offsetof(s3r_prof_record, launches) == 12 &&
               offsetof(s3r_prof_record, flops) == 16 && offsetof(s3r_prof_record, bytes) == 24 &&
               offsetof(s3r_prof_record, exec_flops) == 32 && offsetof(s3r_prof_record, algo) == 40 &&
               offsetof(s3r_prof_record, reserved) == 44, "s3r_prof_record field offsets");
_Static_assert(S3R_OK == 0 && S3R_ERR_INVALID == -1 && S3R_ERR_HIP == -2 && S3R_ERR_WORKSPACE == -3, "status codes");
_Static_assert(S3R_OP_CONV == 0 && S3R_OP_DECONV == 1 && S3R_OP_LINEAR == 2, "op codes");
_Static_assert(S3R_LAYOUT_PLAIN == 0 && S3R_LAYOUT_WINO_H == 2 && S3R_LAYOUT_WINO_DH == 3 && S3R_LAYOUT_WINO_HW == 4, "layouts");
_Static_assert(S3R_ALGO_AUTO == 0 && S3R_ALGO_DIRECT == 1 && S3R_ALGO_WINOGRAD == 2, "algorithms");

static int parse_desc(char* line, s3r_conv_desc* d) {
    int32_t* f = (int32_t*)d;
    char* tok = strtok(line, " \t\n");          /* the "desc" keyword */
    for (int i = 0; i < 21; ++i) {
        tok = strtok(NULL, " \t\n");
        if (!tok) return -1;
        f[i] = (int32_t)strtol(tok, NULL, 10);
    }
    tok = strtok(NULL, " \t\n");
    if (!tok) return -1;
    d->act_param = strtof(tok, NULL);
    return 0;
}

int main(void) {
    printf("layout abi=%d desc=%zu layer=%zu prof=%zu algo_off=%zu act_param_off=%zu packed_w_off=%zu\n", s3r_abi_version(),
           sizeof(s3r_conv_desc), sizeof(s3r_layer), sizeof(s3r_prof_record), offsetof(s3r_conv_desc, algo),
           offsetof(s3r_conv_desc, act_param), offsetof(s3r_layer, packed_w));
    if (s3r_abi_version() != S3R_ABI_VERSION) {
        fprintf(stderr, "library is ABI %d, header is ABI %d\n", s3r_abi_version(), S3R_ABI_VERSION);
        return 2;
    }
    char line[1024];
    while (fgets(line, sizeof line, stdin)) {
        if (!strncmp(line, "desc", 4)) {
            s3r_conv_desc d;
            if (parse_desc(line, &d)) return 3;
            int64_t packed = -1;
            int rc = s3r_conv_packed_elems(&d, &packed);
            printf("desc %d %d %" PRId64 " %" PRId64 " %d %" PRId64 "\n", s3r_conv_out_size(&d), rc, packed,
                   s3r_conv_scratch_elems(&d), s3r_conv_wino_input_layout(&d), s3r_conv_wino_input_elems(&d));
        } else if (!strncmp(line, "chain", 5)) {
            int n = atoi(line + 5);
            if (n <= 0 || n > 64) return 4;
            s3r_layer* layers = (s3r_layer*)calloc((size_t)n, sizeof(s3r_layer));
            for (int i = 0; i < n; ++i) {
                if (!fgets(line, sizeof line, stdin) || parse_desc(line, &layers[i].desc)) return 5;
            }
            int64_t ws = s3r_chain_workspace_elems(layers, n);
            if (ws < 0) printf("chain %" PRId64 " %s\n", ws, s3r_last_error());
            else printf("chain %" PRId64 "\n", ws);
            free(layers);
        } else if (!strncmp(line, "linear", 6)) {
            int b, ci, co;
            if (sscanf(line + 6, "%d %d %d", &b, &ci, &co) != 3) return 6;
            printf("linear %" PRId64 "\n", s3r_linear_scratch_elems(b, ci, co));
        }
    }
    return 0;
}
