// Trained-model PLY <-> device layout (the on-disk format on the input side of the
// path at inference, reference scene/gaussian_model.py:316-331 attribute order, :333-375 save_ply,
// :377-459 load_ply; read through plyfile there, scene/__init__.py:48-66).
//
// File: "ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... /
// end_header", then N records of float32: x y z nx ny nz f_dc_0..2 f_rest_0..(3(K-1)-1) opacity
// scale_0..2 rot_0..3.  f_rest is CHANNEL-major on disk (save_ply transposes (N,K-1,3) -> (N,3,K-1));
// load_ply undoes it.  The header is parsed on the host (any property order, extra float/int/uchar
// properties skipped by size); the records are uploaded as they are and unpacked on the GPU:
// a wave stages its 64 consecutive records in LDS with coalesced dword loads and every output array is
// then written as one contiguous run per wave.  Pure data movement for the raw parameters (bit-exact);
// optionally also the activated tensors the rasterizer takes (cat / sigmoid / exp / normalize,
// reference :104-142).  HBM-bound: record bytes in, the same bytes out.
#include "gsplat_common.h"
#include "../../include/gsplat.h"

#include <stdlib.h>
#include <string.h>

namespace {

constexpr int PLY_THREADS = 256;
constexpr int PLY_MAX_STRIDE_DW = 128;   // 512-byte records at most (degree-3 model: 62 dwords)

struct PlyDev {
    int stride_dw;
    int rest;                      // number of f_rest properties (3 (K-1))
    int off[14];                   // dword offsets: xyz 0-2, f_dc 3-5, opacity 6, scale 7-9, rot 10-13
    int off_rest[GS_PLY_MAX_REST]; // dword offsets of f_rest_j in file order
};

// RT: the f_rest count when known at compile time (0 / 9 / 24 / 45 = degree 0..3: the divisions by the row widths
// become multiply-shifts), -1 = read it from the layout.  CONTIG: f_rest_j sits at off_rest[0] + j (every file the
// reference writes): no offset table lookups.
template <bool PACK, int RT, bool CONTIG>
__global__ __launch_bounds__(PLY_THREADS) void ply_kernel(long long N, PlyDev L, bool aligned16, uint32_t* __restrict__ records,
                                                           float* __restrict__ xyz, float* __restrict__ f_dc,
                                                           float* __restrict__ f_rest, float* __restrict__ opacity,
                                                           float* __restrict__ scaling, float* __restrict__ rotation,
                                                           float* __restrict__ shs, float* __restrict__ opacity_act,
                                                           float* __restrict__ scales_act,
                                                           float* __restrict__ rotations_act) {
    extern __shared__ uint32_t lds[];
    __shared__ int s_off_rest[GS_PLY_MAX_REST];
    for (int j = threadIdx.x; j < L.rest; j += blockDim.x) s_off_rest[j] = L.off_rest[j];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i0 = ((long long)blockIdx.x * (blockDim.x / 64) + wave) * 64;   // first Gaussian of this wave
    if (i0 >= N) return;
    const int cnt = (int)(N - i0 < 64 ? N - i0 : 64);
    const int S = L.stride_dw;
    uint32_t* rec = lds + wave * 64 * S;
    uint32_t* g_rec = records + i0 * S;
    const int R = RT >= 0 ? RT : L.rest, coeffs = R / 3;
    const int Rd = (RT >= 0 ? RT : L.rest) > 0 ? (RT >= 0 ? RT : L.rest) : 1;   // divisor (no f_rest: loops are empty)
    const int rest0 = L.off_rest[0];
    auto F = [&](int gi, int off_dw) -> uint32_t& { return rec[gi * S + off_dw]; };
    auto rest_at = [&](int k, int c) -> int {     // parameter (k, c) <- file column c * coeffs + k
        return CONTIG ? rest0 + c * coeffs + k : s_off_rest[c * coeffs + k];
    };
    if (!PACK) {
        {   // 64 records = 256 S bytes, a multiple of 16: 16-byte loads, eight in flight per lane
            const int n4 = (cnt * S) / 4;
            const uint4* g4 = reinterpret_cast<const uint4*>(g_rec);
            uint4* r4 = reinterpret_cast<uint4*>(rec);
            int m = lane;
            if (aligned16) {
                for (; m + 7 * 64 < n4; m += 8 * 64) {
                    uint4 t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = g4[m + 64 * k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) r4[m + 64 * k] = t[k];
                }
                for (; m < n4; m += 64) r4[m] = g4[m];
                for (int t = 4 * n4 + lane; t < cnt * S; t += 64) rec[t] = g_rec[t];
            } else {
                for (; m < cnt * S; m += 64) rec[m] = g_rec[m];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // every output: one contiguous run of this wave's Gaussians, lane m -> (Gaussian m / width, column m % width)
        if (xyz) for (int m = lane; m < cnt * 3; m += 64) xyz[i0 * 3 + m] = __uint_as_float(F(m / 3, L.off[m % 3]));
        if (f_dc) for (int m = lane; m < cnt * 3; m += 64) f_dc[i0 * 3 + m] = __uint_as_float(F(m / 3, L.off[3 + m % 3]));
        if (f_rest)
#pragma unroll 4
            for (int m = lane; m < cnt * R; m += 64) {
                const int gi = m / Rd, j = m - gi * Rd;
                f_rest[i0 * R + m] = __uint_as_float(F(gi, rest_at(j / 3, j % 3)));
            }
        if (opacity && lane < cnt) opacity[i0 + lane] = __uint_as_float(F(lane, L.off[6]));
        if (scaling) for (int m = lane; m < cnt * 3; m += 64) scaling[i0 * 3 + m] = __uint_as_float(F(m / 3, L.off[7 + m % 3]));
        if (rotation) for (int m = lane; m < cnt * 4; m += 64) rotation[i0 * 4 + m] = __uint_as_float(F(m / 4, L.off[10 + m % 4]));
        // ---- what the rasterizer takes (reference :104-142): shs = cat(f_dc, f_rest), sigmoid, exp, normalize
        if (shs) {
            const int RL = R + 3;
#pragma unroll 4
            for (int m = lane; m < cnt * RL; m += 64) {
                const int gi = m / RL, j = m - gi * RL, k = j / 3, c = j % 3;
                shs[i0 * RL + m] = __uint_as_float(k == 0 ? F(gi, L.off[3 + c]) : F(gi, rest_at(k - 1, c)));
            }
        }
        if (opacity_act && lane < cnt) opacity_act[i0 + lane] = 1.0f / (1.0f + expf(-__uint_as_float(F(lane, L.off[6]))));
        if (scales_act) for (int m = lane; m < cnt * 3; m += 64) scales_act[i0 * 3 + m] = expf(__uint_as_float(F(m / 3, L.off[7 + m % 3])));
        if (rotations_act)
            for (int m = lane; m < cnt * 4; m += 64) {
                const int gi = m / 4;
                const float a = __uint_as_float(F(gi, L.off[10])), b = __uint_as_float(F(gi, L.off[11])),
                            c = __uint_as_float(F(gi, L.off[12])), d = __uint_as_float(F(gi, L.off[13]));
                const float n = fmaxf(sqrtf(a * a + b * b + c * c + d * d), 1e-12f);   // F.normalize: x / max(|x|, eps)
                rotations_act[i0 * 4 + m] = __uint_as_float(F(gi, L.off[10 + m % 4])) / n;
            }
    } else {
        // save_ply: build the records in LDS (normals and unknown columns zero), then one contiguous store
        for (int m = lane; m < cnt * S; m += 64) rec[m] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int m = lane; m < cnt * 3; m += 64) F(m / 3, L.off[m % 3]) = __float_as_uint(xyz[i0 * 3 + m]);
        for (int m = lane; m < cnt * 3; m += 64) F(m / 3, L.off[3 + m % 3]) = __float_as_uint(f_dc[i0 * 3 + m]);
        for (int m = lane; m < cnt * R; m += 64) {
            const int gi = m / Rd, j = m - gi * Rd;
            F(gi, rest_at(j / 3, j % 3)) = __float_as_uint(f_rest[i0 * R + m]);
        }
        if (lane < cnt) F(lane, L.off[6]) = __float_as_uint(opacity[i0 + lane]);
        for (int m = lane; m < cnt * 3; m += 64) F(m / 3, L.off[7 + m % 3]) = __float_as_uint(scaling[i0 * 3 + m]);
        for (int m = lane; m < cnt * 4; m += 64) F(m / 4, L.off[10 + m % 4]) = __float_as_uint(rotation[i0 * 4 + m]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int m = lane; m < cnt * S; m += 64) g_rec[m] = rec[m];
    }
}

int type_size(const char* t) {
    static const struct { const char* name; int size; } T[] = {
        {"char", 1}, {"uchar", 1}, {"int8", 1}, {"uint8", 1}, {"short", 2}, {"ushort", 2}, {"int16", 2}, {"uint16", 2},
        {"int", 4}, {"uint", 4}, {"int32", 4}, {"uint32", 4}, {"float", 4}, {"float32", 4}, {"double", 8}, {"float64", 8}};
    for (const auto& e : T)
        if (!strcmp(t, e.name)) return e.size;
    return 0;
}

bool is_f32(const char* t) { return !strcmp(t, "float") || !strcmp(t, "float32"); }

int make_dev_layout(const GsPlyLayout* L, PlyDev* D) {
    GS_REQUIRE(L->stride_bytes > 0 && L->stride_bytes % 4 == 0 && L->stride_bytes / 4 <= PLY_MAX_STRIDE_DW,
               "record stride %d bytes unsupported (multiple of 4, at most %d)", L->stride_bytes, 4 * PLY_MAX_STRIDE_DW);
    GS_REQUIRE(L->sh_rest_count >= 0 && L->sh_rest_count <= GS_PLY_MAX_REST && L->sh_rest_count % 3 == 0,
               "f_rest count %d unsupported", L->sh_rest_count);
    D->stride_dw = L->stride_bytes / 4;
    D->rest = L->sh_rest_count;
    const int* src[14] = {&L->off_xyz[0], &L->off_xyz[1], &L->off_xyz[2], &L->off_f_dc[0], &L->off_f_dc[1], &L->off_f_dc[2],
                          &L->off_opacity, &L->off_scale[0], &L->off_scale[1], &L->off_scale[2], &L->off_rot[0],
                          &L->off_rot[1], &L->off_rot[2], &L->off_rot[3]};
    for (int k = 0; k < 14; ++k) {
        GS_REQUIRE(*src[k] >= 0 && *src[k] % 4 == 0 && *src[k] + 4 <= L->stride_bytes, "bad field offset %d", *src[k]);
        D->off[k] = *src[k] / 4;
    }
    for (int j = 0; j < L->sh_rest_count; ++j) {
        GS_REQUIRE(L->off_f_rest[j] >= 0 && L->off_f_rest[j] % 4 == 0 && L->off_f_rest[j] + 4 <= L->stride_bytes,
                   "bad f_rest offset %d", L->off_f_rest[j]);
        D->off_rest[j] = L->off_f_rest[j] / 4;
    }
    return GS_OK;
}

}  // namespace

extern "C" int gs_ply_parse_header(const void* file_bytes, size_t n_bytes, size_t file_size, GsPlyLayout* out) {
    GS_REQUIRE(file_bytes && out, "null argument");
    const char* p = (const char*)file_bytes;
    const char* end = p + n_bytes;
    memset(out, 0, sizeof(*out));
    for (int k = 0; k < 3; ++k) out->off_xyz[k] = out->off_f_dc[k] = out->off_scale[k] = -1;
    for (int k = 0; k < 4; ++k) out->off_rot[k] = -1;
    out->off_opacity = -1;
    bool in_vertex = false, seen_vertex = false, seen_format = false, done = false;
    int line_no = 0, offset = 0, nscale = 0, nrot = 0;
    while (p < end && !done) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        GS_REQUIRE(nl, "PLY header is not terminated by end_header");
        char line[256];
        size_t len = (size_t)(nl - p);
        if (len > 0 && p[len - 1] == '\r') --len;
        GS_REQUIRE(len < sizeof(line), "PLY header line %d too long", line_no + 1);
        memcpy(line, p, len);
        line[len] = 0;
        p = nl + 1;
        ++line_no;
        if (line_no == 1) {
            GS_REQUIRE(!strcmp(line, "ply"), "not a PLY file (magic)");
            continue;
        }
        char a[64] = "", b[64] = "", c[64] = "";
        const int nf = sscanf(line, "%63s %63s %63s", a, b, c);
        if (nf <= 0 || !strcmp(a, "comment") || !strcmp(a, "obj_info")) continue;
        if (!strcmp(a, "format")) {
            GS_REQUIRE(nf >= 2 && !strcmp(b, "binary_little_endian"), "PLY format '%s' unsupported (binary_little_endian only)", b);
            seen_format = true;
        } else if (!strcmp(a, "element")) {
            GS_REQUIRE(nf == 3, "bad element line");
            if (!strcmp(b, "vertex")) {
                GS_REQUIRE(!seen_vertex, "two vertex elements");
                in_vertex = seen_vertex = true;
                out->num_vertices = atoll(c);
                GS_REQUIRE(out->num_vertices >= 0, "negative vertex count");
            } else {
                GS_REQUIRE(seen_vertex, "element '%s' precedes the vertex element: unsupported", b);
                in_vertex = false;   // elements after the vertices are not read
            }
        } else if (!strcmp(a, "property")) {
            if (!in_vertex) continue;
            GS_REQUIRE(nf == 3 && strcmp(b, "list") != 0, "list / malformed property in the vertex element");
            const int sz = type_size(b);
            GS_REQUIRE(sz > 0, "unknown property type '%s'", b);
            int* slot = nullptr;
            if (!strcmp(c, "x")) slot = &out->off_xyz[0];
            else if (!strcmp(c, "y")) slot = &out->off_xyz[1];
            else if (!strcmp(c, "z")) slot = &out->off_xyz[2];
            else if (!strcmp(c, "opacity")) slot = &out->off_opacity;
            else if (!strncmp(c, "f_dc_", 5)) {
                const int k = atoi(c + 5);
                GS_REQUIRE(k >= 0 && k < 3, "unexpected property %s", c);
                slot = &out->off_f_dc[k];
            } else if (!strncmp(c, "f_rest_", 7)) {      // taken in FILE order, as load_ply does (:399-408)
                GS_REQUIRE(out->sh_rest_count < GS_PLY_MAX_REST, "more than %d f_rest properties", GS_PLY_MAX_REST);
                slot = &out->off_f_rest[out->sh_rest_count++];
            } else if (!strncmp(c, "scale_", 6)) {
                GS_REQUIRE(nscale < 3, "more than 3 scale properties");
                slot = &out->off_scale[nscale++];
            } else if (!strncmp(c, "rot", 3)) {
                GS_REQUIRE(nrot < 4, "more than 4 rot properties");
                slot = &out->off_rot[nrot++];
            }
            if (slot) {
                GS_REQUIRE(is_f32(b), "property %s has type %s: float expected", c, b);
                *slot = offset;
            }
            offset += sz;
        } else if (!strcmp(a, "end_header")) {
            done = true;
        }
    }
    GS_REQUIRE(done, "PLY header is not terminated by end_header");
    GS_REQUIRE(seen_format && seen_vertex, "PLY header lacks a format or a vertex element");
    out->stride_bytes = offset;
    out->data_offset = (long long)(p - (const char*)file_bytes);
    const bool all = out->off_xyz[0] >= 0 && out->off_xyz[1] >= 0 && out->off_xyz[2] >= 0 && out->off_f_dc[0] >= 0 &&
                     out->off_f_dc[1] >= 0 && out->off_f_dc[2] >= 0 && out->off_opacity >= 0 && nscale == 3 && nrot == 4;
    GS_REQUIRE(all, "PLY lacks one of x y z f_dc_0..2 opacity scale_0..2 rot_0..3");
    GS_REQUIRE(out->sh_rest_count % 3 == 0, "f_rest count %d is not a multiple of 3", out->sh_rest_count);
    GS_REQUIRE(out->data_offset + out->num_vertices * (long long)out->stride_bytes <= (long long)file_size,
               "file too short: %lld vertices of %d bytes after a %lld-byte header, file has %zu bytes", out->num_vertices,
               out->stride_bytes, out->data_offset, file_size);
    return GS_OK;
}

extern "C" long long gs_ply_write_header(char* buf, size_t cap, long long num_vertices, int sh_rest_count, GsPlyLayout* layout) {
    // the header plyfile writes for save_ply's element (:358-375): one "property float" per attribute of :316-331
    char* tmp = (char*)malloc(4096 + 32 * (size_t)(sh_rest_count > 0 ? sh_rest_count : 0));
    if (!tmp) return -1;
    int n = sprintf(tmp, "ply\nformat binary_little_endian 1.0\nelement vertex %lld\n", num_vertices);
    GsPlyLayout L;
    memset(&L, 0, sizeof(L));
    int off = 0;
    auto prop = [&](const char* name, int* slot) {
        n += sprintf(tmp + n, "property float %s\n", name);
        if (slot) *slot = off;
        off += 4;
    };
    char name[32];
    prop("x", &L.off_xyz[0]); prop("y", &L.off_xyz[1]); prop("z", &L.off_xyz[2]);
    prop("nx", nullptr); prop("ny", nullptr); prop("nz", nullptr);
    for (int k = 0; k < 3; ++k) { sprintf(name, "f_dc_%d", k); prop(name, &L.off_f_dc[k]); }
    for (int k = 0; k < sh_rest_count && k < GS_PLY_MAX_REST; ++k) { sprintf(name, "f_rest_%d", k); prop(name, &L.off_f_rest[k]); }
    prop("opacity", &L.off_opacity);
    for (int k = 0; k < 3; ++k) { sprintf(name, "scale_%d", k); prop(name, &L.off_scale[k]); }
    for (int k = 0; k < 4; ++k) { sprintf(name, "rot_%d", k); prop(name, &L.off_rot[k]); }
    n += sprintf(tmp + n, "end_header\n");
    L.num_vertices = num_vertices;
    L.data_offset = n;
    L.stride_bytes = off;
    L.sh_rest_count = sh_rest_count;
    if (layout) *layout = L;
    if (buf && (size_t)n <= cap) memcpy(buf, tmp, (size_t)n);
    free(tmp);
    return n;
}

static int ply_launch(bool pack, const GsPlyLayout* layout, void* records, float* xyz, float* f_dc, float* f_rest,
                      float* opacity, float* scaling, float* rotation, float* shs, float* opacity_act,
                      float* scales_act, float* rotations_act, void* stream_) {
    GS_REQUIRE(layout, "layout is null");
    PlyDev D;
    int rc = make_dev_layout(layout, &D);
    if (rc) return rc;
    const long long N = layout->num_vertices;
    if (N == 0) return GS_OK;
    GS_REQUIRE(records && ((uintptr_t)records & 3) == 0, "records must be a 4-byte aligned device pointer");
    GS_REQUIRE(!pack || (xyz && f_dc && opacity && scaling && rotation && (f_rest || D.rest == 0)), "pack needs every parameter");
    // waves per workgroup: as many as keep the staged records within 64 KB of LDS (4 for a degree-3 model)
    // 64 KB of LDS per workgroup: the dynamic staging area shares it with the static offset table (s_off_rest)
    int wpb = (int)((65536 - sizeof(int) * GS_PLY_MAX_REST) / ((size_t)64 * D.stride_dw * 4));
    wpb = wpb > PLY_THREADS / 64 ? PLY_THREADS / 64 : wpb;
    const size_t lds_bytes = (size_t)wpb * 64 * D.stride_dw * 4;
    const long long waves = (N + 63) / 64;
    const dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
    hipStream_t stream = (hipStream_t)stream_;
    const bool aligned16 = ((uintptr_t)records & 15) == 0;
    bool contig = true;
    for (int j = 1; j < D.rest; ++j) contig = contig && D.off_rest[j] == D.off_rest[0] + j;
#define GS_PLY_LAUNCH(PACK_, RT_, CONTIG_)                                                                          \
    hipLaunchKernelGGL((ply_kernel<PACK_, RT_, CONTIG_>), grid, block, lds_bytes, stream, N, D, aligned16,           \
                       (uint32_t*)records, xyz, f_dc, f_rest, opacity, scaling, rotation, shs, opacity_act,          \
                       scales_act, rotations_act)
#define GS_PLY_DISPATCH(PACK_)                                                   \
    do {                                                                         \
        if (contig && D.rest == 45) GS_PLY_LAUNCH(PACK_, 45, true);              \
        else if (contig && D.rest == 24) GS_PLY_LAUNCH(PACK_, 24, true);         \
        else if (contig && D.rest == 9) GS_PLY_LAUNCH(PACK_, 9, true);           \
        else if (D.rest == 0) GS_PLY_LAUNCH(PACK_, 0, true);                     \
        else GS_PLY_LAUNCH(PACK_, -1, false);                                    \
    } while (0)
    if (pack) GS_PLY_DISPATCH(true);
    else GS_PLY_DISPATCH(false);
#undef GS_PLY_DISPATCH
#undef GS_PLY_LAUNCH
    GS_CHECK_LAUNCH(pack ? "ply_pack" : "ply_unpack");
    return GS_OK;
}

extern "C" int gs_ply_unpack(const GsPlyLayout* layout, const void* records, float* xyz, float* features_dc,
                             float* features_rest, float* opacity, float* scaling, float* rotation, float* shs,
                             float* opacities_act, float* scales_act, float* rotations_act, void* stream) {
    return ply_launch(false, layout, (void*)records, xyz, features_dc, features_rest, opacity, scaling, rotation, shs,
                      opacities_act, scales_act, rotations_act, stream);
}

extern "C" int gs_ply_pack(const GsPlyLayout* layout, void* records, const float* xyz, const float* features_dc,
                           const float* features_rest, const float* opacity, const float* scaling,
                           const float* rotation, void* stream) {
    return ply_launch(true, layout, records, (float*)xyz, (float*)features_dc, (float*)features_rest, (float*)opacity,
                      (float*)scaling, (float*)rotation, nullptr, nullptr, nullptr, nullptr, stream);
}
