/* The GPU leg of the compiled-C consumer: one s3r_conv_forward and one s3r_chamfer_forward driven from plain C through
 * include/s3r.h, device memory from libamdhip64 resolved with dlopen (no HIP headers: a C / cgo / JNI maintainer needs
 * only hipMalloc / hipMemcpy / hipMemset / hipDeviceSynchronize / hipFree).  tests/test_c_abi_gpu.py writes the inputs as raw
 * little-endian files, runs this, and compares the outputs with tests/golden/.
 *
 *   consumer_gpu conv <dir> <22 desc fields>    reads x.bin (halo-padded input), w.bin (torch layout), scale.bin, shift.bin;
 *                                               packs the weights on the device, runs the layer, writes y.bin
 *   consumer_gpu chamfer <dir> <B> <N> <M>      reads p.bin, q.bin; writes d1.bin d2.bin i1.bin i2.bin
 */
#include <dlfcn.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "s3r.h"

typedef int (*malloc_fn)(void**, size_t);
typedef int (*free_fn)(void*);
typedef int (*memcpy_fn)(void*, const void*, size_t, int);
typedef int (*memset_fn)(void*, int, size_t);
typedef int (*sync_fn)(void);
static malloc_fn hip_malloc;
static free_fn hip_free;
static memcpy_fn hip_memcpy;
static memset_fn hip_memset;
static sync_fn hip_sync;
enum { H2D = 1, D2H = 2 };

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); exit(1); } while (0)
#define HIP(call) do { int e_ = (call); if (e_) DIE("%s -> hip error %d", #call, e_); } while (0)
#define S3R(call) do { int e_ = (call); if (e_ < 0) DIE("%s -> %d: %s", #call, e_, s3r_last_error()); } while (0)

static void* read_file(const char* dir, const char* name, size_t* bytes) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) DIE("cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)n);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) DIE("short read of %s", path);
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

static void write_file(const char* dir, const char* name, const void* p, size_t bytes) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) DIE("cannot write %s", path);
    fclose(f);
}

static void* to_device(const char* dir, const char* name, size_t* bytes) {
    void* h = read_file(dir, name, bytes);
    void* d = NULL;
    HIP(hip_malloc(&d, *bytes));
    HIP(hip_memcpy(d, h, *bytes, H2D));
    free(h);
    return d;
}

static void from_device(const char* dir, const char* name, const void* d, size_t bytes) {
    void* h = malloc(bytes);
    HIP(hip_memcpy(h, d, bytes, D2H));
    write_file(dir, name, h, bytes);
    free(h);
}

static int64_t ipow(int64_t b, int e) {
    int64_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}

int main(int argc, char** argv) {
    void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hip) DIE("dlopen libamdhip64.so: %s", dlerror());
    hip_malloc = (malloc_fn)dlsym(hip, "hipMalloc");
    hip_free = (free_fn)dlsym(hip, "hipFree");
    hip_memcpy = (memcpy_fn)dlsym(hip, "hipMemcpy");
    hip_memset = (memset_fn)dlsym(hip, "hipMemset");
    hip_sync = (sync_fn)dlsym(hip, "hipDeviceSynchronize");
    if (!hip_malloc || !hip_free || !hip_memcpy || !hip_memset || !hip_sync) DIE("libamdhip64 lacks a runtime symbol");
    if (s3r_abi_version() != S3R_ABI_VERSION) DIE("library ABI %d != header ABI %d", s3r_abi_version(), S3R_ABI_VERSION);
    if (argc < 3) DIE("usage: consumer_gpu conv|chamfer <dir> ...");
    const char* dir = argv[2];

    if (!strcmp(argv[1], "conv")) {
        if (argc != 3 + 22) DIE("conv needs 22 descriptor fields");
        s3r_conv_desc d;
        int32_t* f = (int32_t*)&d;
        for (int i = 0; i < 21; ++i) f[i] = (int32_t)strtol(argv[3 + i], NULL, 10);
        d.act_param = strtof(argv[3 + 21], NULL);
        size_t nx, nw, ns, nb;
        void* x = to_device(dir, "x.bin", &nx);
        void* w = to_device(dir, "w.bin", &nw);
        float* scale = (float*)to_device(dir, "scale.bin", &ns);
        float* shift = (float*)to_device(dir, "shift.bin", &nb);
        int64_t packed_elems = 0;
        S3R(s3r_conv_packed_elems(&d, &packed_elems));
        void* packed = NULL;
        HIP(hip_malloc(&packed, (size_t)packed_elems * 4));
        S3R(s3r_conv_pack_weights(&d, (const float*)w, packed, NULL));
        int64_t scratch_elems = s3r_conv_scratch_elems(&d);
        if (scratch_elems < 0) DIE("scratch query: %s", s3r_last_error());
        float* scratch = NULL;
        if (scratch_elems) {
            HIP(hip_malloc((void**)&scratch, (size_t)scratch_elems * 4));
            HIP(hip_memset(scratch, 0, (size_t)scratch_elems * 4));
        }
        int m = s3r_conv_out_size(&d);
        int64_t ny = (int64_t)d.batch * d.cout * ipow(m + 2 * d.out_halo, d.ndim);
        void* y = NULL;
        HIP(hip_malloc(&y, (size_t)ny * 4));
        HIP(hip_memset(y, 0, (size_t)ny * 4));
        S3R(s3r_conv_forward(&d, x, packed, scale, shift, y, scratch, scratch_elems, NULL));
        HIP(hip_sync());
        from_device(dir, "y.bin", y, (size_t)ny * 4);
        printf("conv out_size=%d packed=%" PRId64 " scratch=%" PRId64 " y_elems=%" PRId64 "\n", m, packed_elems, scratch_elems, ny);
        hip_free(x);
        hip_free(w);
        hip_free(scale);
        hip_free(shift);
        hip_free(packed);
        hip_free(y);
        if (scratch) hip_free(scratch);
    } else if (!strcmp(argv[1], "chamfer")) {
        if (argc != 6) DIE("chamfer needs B N M");
        int B = atoi(argv[3]), N = atoi(argv[4]), M = atoi(argv[5]);
        size_t np, nq;
        float* p = (float*)to_device(dir, "p.bin", &np);
        float* q = (float*)to_device(dir, "q.bin", &nq);
        if (np != (size_t)B * N * 12 || nq != (size_t)B * M * 12) DIE("cloud sizes do not match B N M");
        float *d1, *d2;
        int32_t *i1, *i2;
        HIP(hip_malloc((void**)&d1, (size_t)B * N * 4));
        HIP(hip_malloc((void**)&i1, (size_t)B * N * 4));
        HIP(hip_malloc((void**)&d2, (size_t)B * M * 4));
        HIP(hip_malloc((void**)&i2, (size_t)B * M * 4));
        S3R(s3r_chamfer_forward(p, q, d1, d2, i1, i2, B, N, M, NULL));
        HIP(hip_sync());
        from_device(dir, "d1.bin", d1, (size_t)B * N * 4);
        from_device(dir, "i1.bin", i1, (size_t)B * N * 4);
        from_device(dir, "d2.bin", d2, (size_t)B * M * 4);
        from_device(dir, "i2.bin", i2, (size_t)B * M * 4);
        printf("chamfer B=%d N=%d M=%d\n", B, N, M);
        hip_free(p);
        hip_free(q);
        hip_free(d1);
        hip_free(d2);
        hip_free(i1);
        hip_free(i2);
    } else {
        DIE("unknown mode %s", argv[1]);
    }
    return 0;
}
