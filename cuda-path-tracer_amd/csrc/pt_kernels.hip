/*
 * pt_kernels.hip -- closest-hit query kernel, primary-ray kernel, the persistent path-tracing
 * kernel and the split-buffer compounding kernels (gfx950).
 *
 * Render kernel structure: a persistent grid (a few 256-thread workgroups per CU) whose waves pull
 * work items {8x8-pixel tile, sample slice} from per-workgroup queues (contiguous ranges of the item
 * space in macro-tile order; an empty queue steals from the next).  A lane owns one pixel of an
 * item and runs the slice's samples one after the other; every trip of the main loop advances each
 * lane's path by ONE segment (closest hit + shading).  A lane whose path ended starts its next
 * sample on the next trip, and a lane whose slice is finished immediately takes the next free
 * pixel of the wave's current item (or of a freshly pulled one), so terminated paths never leave
 * holes in the wave: the lanes stay compacted onto live rays until the queues run dry.  Each
 * (pixel, slice) sum is written exactly once by its owner to a partial buffer and the partials are
 * combined in fixed order afterwards -- no atomics on pixels, bit-reproducible images.
 *
 * The kernel is instantiated per register budget (waves per SIMD) and per leaf strategy: lane by lane
 * (leaf_block) for small scenes, or the cooperative phase (leaf_coop: the waiting triangle runs of a
 * wave are tested by all of its lanes) for scenes whose time goes into the octree.
 */
#include <algorithm>
#include <cstdlib>

#include "pt_device.h"
#include "pt_internal.h"

namespace ptamd {

static constexpr int kBlock = 256;
/* frames of the per-lane octree stack: one 16-byte and one 4-byte word per level, each level an array over the
 * workgroup's lanes; the walk pushes at levels 0 .. depth-2 (children of level depth-1 nodes are leaves) */
__host__ __device__ static inline int stack_levels(int max_depth) { return max_depth > 1 ? max_depth - 1 : 1; }

/* the node step's look-up tables (pt_child_table.h), made at compile time; every workgroup copies them to the start of its LDS */
__device__ const ChildTables kChildTablesDev = make_child_tables();
static constexpr int kTableWords = (int)(sizeof(ChildTables) / 4);
static constexpr int kEntityQuads = (int)(kLdsEntities * sizeof(PtEntity) / 16);
/* 16-byte words of LDS in front of the stack frames: the tables, then room for the entity records of a small scene */
static constexpr int kConstQuads = (int)(sizeof(ChildTables) / 16) + kEntityQuads;
static_assert(sizeof(ChildTables) % 16 == 0 && sizeof(PtEntity) % 16 == 0, "the stack frames behind the constants are 16-byte words");
static_assert(sizeof(ChildTables) == kEntLdsByte, "entity_block addresses the entity records by this offset");

/* Fills the constant part of a workgroup's LDS; returns whether the scene's entity records are in it (entity_block). */
__device__ __forceinline__ bool load_lds_constants(float4 *lds, const PtSceneDev &sc) {
    /* the node step addresses the tables from LDS address 0: the kernels here have no static LDS in front of the dynamic segment */
    if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&kChildTablesDev);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < kTableWords; i += kBlock) dst[i] = src[i];
    const bool ents = kLdsEntities > 0 && sc.n_entities <= kLdsEntities;
    if (ents) {
        const float4 *es = reinterpret_cast<const float4 *>(sc.entities);
        float4 *ed = lds + sizeof(ChildTables) / 16;
        const int n = sc.n_entities * (int)(sizeof(PtEntity) / 16);
        for (int i = threadIdx.x; i < n; i += kBlock) ed[i] = es[i];
    }
    __syncthreads();
    return ents;
}

/* dynamic LDS of a workgroup: the tables and entity records, the stack frames (16-byte words, then 4-byte words), then one
 * CoopLds per wave.  `lds` here is the start of the stack frames. */
__device__ __forceinline__ CoopLds *coop_area(float4 *lds, int max_depth) {
    char *base = reinterpret_cast<char *>(lds) + (size_t)stack_levels(max_depth) * kBlock * (sizeof(float4) + sizeof(uint32_t));
    return reinterpret_cast<CoopLds *>(base) + (threadIdx.x >> 6);
}

/* ------------------------------------------------------------------ shading (fast math allowed) */
/* The library is built with -ffp-contract=off because geometry must round like a scalar IEEE evaluation.  Shading is
 * compared with the reference by tolerance, so its functions fuse a multiply with the add it feeds WITHIN one expression
 * (contract(on): decided by the front end from the source text alone.  contract(fast) lets the optimiser fuse across
 * statements depending on how often a product is used, which differs between the instantiations of the render kernel:
 * their images must be bit-identical). */
#define PT_SHADING_FP _Pragma("clang fp contract(on)")

__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ V3 fast_normalized(V3 v) {
    PT_SHADING_FP
    float r = fast_rsq(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk(v.x * r, v.y * r, v.z * r);
}
/* x^y for x in (0,1] (and 1000^y): v_log_f32 / v_exp_f32, like the reference's __powf */
__device__ __forceinline__ float fast_pow(float x, float y) {
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
/* radiance^contrast (src/scene.cuh:78-80 calls powf): radiance is never negative; the library routine costs ~230
 * instructions per call, v_log_f32 / v_exp_f32 give 1e-6 relative accuracy */
__device__ __forceinline__ float hdri_pow(float x, float y) {
    if (x > 0.0f) return fast_pow(x, y);
    return y == 0.0f ? 1.0f : (y > 0.0f ? 0.0f : __builtin_inff());
}
/* acos(x), |x| <= 1 (values just above 1, a rounding of the normalisation in front of it, give 0 or pi): the polynomial
 * of Abramowitz & Stegun 4.4.46, sqrt(1 - x) * P7(x) with |error| <= 2e-8 before rounding, mirrored for negative x.  Twelve
 * instructions; the library's acosf is 130, and it is evaluated by every wave in every shading round for the few lanes
 * whose ray left the scene.  The result moves the environment look-up by 1e-4 of a texel at most. */
__device__ __forceinline__ float fast_acos(float x) {
    PT_SHADING_FP
    const float a = fabsf(x);
    const float s = __builtin_amdgcn_sqrtf(fmaxf(1.0f - a, 0.0f));
    float p = -0.0012624911f;
    p = p * a + 0.0066700901f;
    p = p * a - 0.0170881256f;
    p = p * a + 0.0308918810f;
    p = p * a - 0.0501743046f;
    p = p * a + 0.0889789874f;
    p = p * a - 0.2145988016f;
    p = p * a + 1.5707963050f;
    const float r = s * p;
    return (__float_as_uint(x) >> 31) ? 3.14159265f - r : r;
}
/* sin / cos of 2*pi*rev, rev in (0,1]: v_sin_f32 / v_cos_f32 take revolutions */
__device__ __forceinline__ float sin_rev(float rev) { return __builtin_amdgcn_sinf(rev); }
__device__ __forceinline__ float cos_rev(float rev) { return __builtin_amdgcn_cosf(rev); }

__device__ __forceinline__ float luminosity(V3 c) {
    PT_SHADING_FP
    return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }  /* src/util.cuh:40 */

/* Traffic probes (profiles/fetch_attr.sh): builds that read texel 0 of every texture (PT_PROBE_TEXEL0) or of the
 * environment map (PT_PROBE_HDRI0) instead of the addressed one -- same instructions, same paths through the scene up to
 * the colours, but the array stays in cache: the drop in memory-side fetch bytes is that array's share.  Not renders. */
__device__ __forceinline__ int to_index(float f, int n) {   /* C truncation, clamped into the image (NaN -> 0) */
    return (int)__builtin_amdgcn_fmed3f(f, 0.0f, (float)(n - 1));      /* n < 2^24: n - 1 is exact; one clamp, no branches */
}

/* src/texture.cuh:22-43 (indices clamped into the texture) */
__device__ __forceinline__ V3 tex_rgb(const uint8_t *data, int w, int h, float u, float v) {
    int x = to_index(u * (float)w, w);
    int y = to_index((1.0f - v) * (float)h, h);
#ifdef PT_PROBE_TEXEL0
    x = 0; y = 0;
#endif
    const __attribute__((address_space(1))) uint8_t *p = (const __attribute__((address_space(1))) uint8_t *)data + 3 * ((size_t)y * (size_t)w + (size_t)x);
    const float k = 1.0f / 255.0f;
    return mk((float)p[0] * k, (float)p[1] * k, (float)p[2] * k);
}
__device__ __forceinline__ float tex_gray(const uint8_t *data, int w, int h, float u, float v) {
    int x = to_index(u * (float)w, w);
    int y = to_index((1.0f - v) * (float)h, h);
#ifdef PT_PROBE_TEXEL0
    x = 0; y = 0;
#endif
    return (float)((const __attribute__((address_space(1))) uint8_t *)data)[(size_t)y * (size_t)w + (size_t)x] * (1.0f / 255.0f);
}

/* The four material samples of one hit (src/material.cuh:21-44: sample_normal, sample_albedo, sample_smoothness,
 * sample_specular; constants where a map is not set).  Maps of one resolution are read from the material's texel
 * records (PtMaterial::packed): one texel index and one 8- or 16-byte load for all of them. */
struct MatSample { V3 albedo, specular, ts_normal; float smoothness; };
template <bool STATS>
__device__ __forceinline__ void sample_material(const PtMaterial &m, float u, float v, MatSample &o, Stats &st) {
    PT_SHADING_FP
    const float k = 1.0f / 255.0f;
    o.ts_normal = mk(0.f, 0.f, 1.f);
    o.albedo = mk(m.albedo[0], m.albedo[1], m.albedo[2]);
    o.smoothness = m.smoothness;
    o.specular = mk(m.specular[0], m.specular[1], m.specular[2]);
    if (m.packed) {
        uint32_t x = (uint32_t)(int)__builtin_amdgcn_fmed3f(u * m.packed_w_f, 0.0f, m.packed_xmax_f);           /* to_index */
        uint32_t y = (uint32_t)(int)__builtin_amdgcn_fmed3f((1.0f - v) * m.packed_h_f, 0.0f, m.packed_ymax_f);
#ifdef PT_PROBE_TEXEL0
        x = 0; y = 0;
#endif
        const uint32_t rec = (((y >> 2) * m.packed_tiles_x + (x >> 2)) << 4) | ((y & 3u) << 2) | (x & 3u);
        /* 32-bit byte offset (checked when packed).  The pointer comes out of the material record, so the compiler cannot tell
         * that it points to global memory: said explicitly, or the texel loads become flat loads */
        typedef const __attribute__((address_space(1))) char *gptr;
        gptr p = (gptr)(reinterpret_cast<const char *>(m.packed)) + (rec << m.packed_shift);
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = *(const __attribute__((address_space(1))) u32x2 *)p;
        if (STATS) st.tex_fetches += (uint32_t)__popc(m.packed_mask);
        if (m.packed_mask & (1u << PT_MAP_NORMAL)) {
            const V3 c = mk((float)(w.x >> 24) * k, (float)(w.y & 0xFFu) * k, (float)((w.y >> 8) & 0xFFu) * k);
            o.ts_normal = fast_normalized(mk(2.0f * c.x - 1.0f, 2.0f * c.y - 1.0f, 2.0f * c.z - 1.0f));
        }
        if (m.packed_mask & (1u << PT_MAP_ALBEDO)) o.albedo = mk((float)(w.x & 0xFFu) * k, (float)((w.x >> 8) & 0xFFu) * k, (float)((w.x >> 16) & 0xFFu) * k);
        if (m.packed_mask & (1u << PT_MAP_SMOOTHNESS)) o.smoothness = (float)((w.y >> 16) & 0xFFu) * k;
        if (m.packed_mask & (1u << PT_MAP_SPECULAR)) {
            const uint32_t w2 = *(const __attribute__((address_space(1))) uint32_t *)(p + 8);
            o.specular = mk((float)(w2 & 0xFFu) * k, (float)((w2 >> 8) & 0xFFu) * k, (float)((w2 >> 16) & 0xFFu) * k);
        }
        return;
    }
    if (m.map[PT_MAP_NORMAL]) {
        if (STATS) st.tex_fetches++;
        const V3 c = tex_rgb(m.map[2], m.map_w[2], m.map_h[2], u, v);
        o.ts_normal = fast_normalized(mk(2.0f * c.x - 1.0f, 2.0f * c.y - 1.0f, 2.0f * c.z - 1.0f));
    }
    if (m.map[PT_MAP_ALBEDO]) { if (STATS) st.tex_fetches++; o.albedo = tex_rgb(m.map[0], m.map_w[0], m.map_h[0], u, v); }
    if (m.map[PT_MAP_SMOOTHNESS]) { if (STATS) st.tex_fetches++; o.smoothness = tex_gray(m.map[3], m.map_w[3], m.map_h[3], u, v); }
    if (m.map[PT_MAP_SPECULAR]) { if (STATS) st.tex_fetches++; o.specular = tex_rgb(m.map[1], m.map_w[1], m.map_h[1], u, v); }
}

/* src/scene.cuh:36-82 */
template <bool STATS>
__device__ __forceinline__ V3 sample_hdri(const PtSceneDev &sc, V3 v, Stats &st) {
    PT_SHADING_FP
    if (!sc.has_hdri) return mk(0.f, 0.f, 0.f);
    if (STATS) st.env_lookups++;
    const float rn = fast_rsq(v.x * v.x + v.z * v.z);
    const float vpx = v.x * rn, vpz = v.z * rn;
    float x_c = ((fast_acos(vpz) * ((__float_as_uint(vpx) >> 31) ? -1.0f : 1.0f)) * (1.0f / (2.0f * 3.14159265f))) + 0.5f;
    const float y_c = fminf(1.0f - ((v.y + 1.0f) * 0.5f), 0.99f);
    x_c += sc.hdri_rot_offset;
    x_c = __builtin_amdgcn_fractf(x_c);      /* fmodf(x, 1) for x >= 0: both are exact */
#ifdef PT_PROBE_HDRI0
    const uint32_t W = 2u, H = 2u;
    const float Wf = 2.0f, Hf = 2.0f, xmax = 1.0f, ymax = 1.0f;
#else
    const uint32_t W = (uint32_t)sc.hdri_w, H = (uint32_t)sc.hdri_h;
    const float Wf = sc.hdri_w_f, Hf = sc.hdri_h_f, xmax = sc.hdri_xmax_f, ymax = sc.hdri_ymax_f;
#endif
    /* texel (x, y): a 32-bit byte offset from the image's base (images of 4 GiB and more are refused when they are set) */
    const char *cols = reinterpret_cast<const char *>(sc.hdri);
#define PT_HDRI_TEXEL(x, y) reinterpret_cast<const float *>(cols + ((y) * W + (x)) * 12u)
    const uint32_t x_l = (uint32_t)(int)__builtin_amdgcn_fmed3f(x_c * Wf, 0.0f, xmax);      /* to_index(x_c * W, W) */
    const uint32_t y_l = (uint32_t)(int)__builtin_amdgcn_fmed3f(y_c * Hf, 0.0f, ymax);
    V3 res;
    if (!sc.use_smooth_hdri) {
        const float *p = PT_HDRI_TEXEL(x_l, y_l);
        res = mk(p[0], p[1], p[2]);
    } else {
        const float xw = fminf(fmaxf(x_c * Wf - (float)(int)x_l, 0.0f), 1.0f);
        const float yw = fminf(fmaxf(y_c * Hf - (float)(int)y_l, 0.0f), 1.0f);
        float ul[3], ur[3], ll[3], lr[3];
        if (sc.hdri_quads) {
            /* the four texels of this look-up, side by side in one 64-byte record */
#ifdef PT_PROBE_HDRI0
            const uint32_t rec = 0u;
#else
            const uint32_t rec = y_l * W + x_l;
#endif
            const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.hdri_quads) + (rec << 6));
            const float4 q0 = q[0], q1 = q[1], q2 = q[2];
            ul[0] = q0.x; ul[1] = q0.y; ul[2] = q0.z; ur[0] = q0.w; ur[1] = q1.x; ur[2] = q1.y;
            ll[0] = q1.z; ll[1] = q1.w; ll[2] = q2.x; lr[0] = q2.y; lr[1] = q2.z; lr[2] = q2.w;
        } else {
            /* (x_l + 1) % W with x_l in [0, W): a wrap, not a division */
            const uint32_t x_r = x_l + 1u == W ? 0u : x_l + 1u, y_r = y_l + 1u == H ? 0u : y_l + 1u;
            const float *a = PT_HDRI_TEXEL(x_l, y_l), *b = PT_HDRI_TEXEL(x_r, y_l), *c = PT_HDRI_TEXEL(x_l, y_r), *d = PT_HDRI_TEXEL(x_r, y_r);
            for (int k = 0; k < 3; k++) { ul[k] = a[k]; ur[k] = b[k]; ll[k] = c[k]; lr[k] = d[k]; }
        }
        /* bilerp of src/math_util.cuh:20-24: lerp(lerp(ul,ll,v), lerp(ur,lr,v), u) */
        float o[3];
        for (int k = 0; k < 3; k++) {
            const float left = (1.0f - yw) * ul[k] + yw * ll[k];
            const float right = (1.0f - yw) * ur[k] + yw * lr[k];
            o[k] = (1.0f - xw) * left + xw * right;
        }
        res = mk(o[0], o[1], o[2]);
    }
#undef PT_HDRI_TEXEL
    if (sc.hdri_contrast != 1.0f) {          /* powf(x, 1) == x */
        res.x = hdri_pow(res.x, sc.hdri_contrast);
        res.y = hdri_pow(res.y, sc.hdri_contrast);
        res.z = hdri_pow(res.z, sc.hdri_contrast);
    }
    return sc.hdri_exposure * res;
}

struct Basis { V3 t, b, n; };
/* src/renderer.cu:168-175 */
__device__ __forceinline__ Basis tangent_space(V3 n) {
    PT_SHADING_FP
    V3 helper = mk(1.f, 0.f, 0.f);
    if (fabsf(n.x) > 0.99f) helper = mk(0.f, 0.f, 1.f);
    Basis B;
    B.t = fast_normalized(cross3(n, helper));
    B.b = fast_normalized(cross3(n, B.t));
    B.n = n;
    return B;
}
/* vec3 * mat3(t,b,n) of src/vector.cuh:186-192 */
__device__ __forceinline__ V3 to_world(V3 v, const Basis &B) {
    PT_SHADING_FP
    return mk(v.x * B.t.x + v.y * B.b.x + v.z * B.n.x, v.x * B.t.y + v.y * B.b.y + v.z * B.n.y,
              v.x * B.t.z + v.y * B.b.z + v.z * B.n.z);
}
/* src/renderer.cu:157-163 */
__device__ __forceinline__ V3 sample_hemisphere(V3 dir, float alpha, float xi_c, float xi_p) {
    PT_SHADING_FP
    const float cos_theta = fast_pow(xi_c, fast_rcp(alpha + 1.0f));
    const float sin_theta = __builtin_amdgcn_sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));      /* v_sqrt_f32: 1 ulp */
    const V3 ts = mk(cos_rev(xi_p) * sin_theta, sin_rev(xi_p) * sin_theta, cos_theta);
    return to_world(ts, tangent_space(dir));
}

/* src/renderer.cu:60-87; xi = Philox block 0 of (pixel, sample); k = make_cam_consts(cam) */
__device__ __forceinline__ void camera_ray(const pt_camera &cam, const CamConsts &k, int u, int v, const float xi[4], V3 &ro, V3 &rd) {
    const V3 pos = mk(cam.position[0], cam.position[1], cam.position[2]);
    const V3 dir = mk(cam.direction[0], cam.direction[1], cam.direction[2]);
    const V3 up = mk(cam.up[0], cam.up[1], cam.up[2]);
    const V3 right = mk(k.right[0], k.right[1], k.right[2]);
    const float n_u = (((float)u + xi[0]) / k.res_x_f) - 0.5f;
    const float n_v = (((float)v + xi[1]) / k.res_y_f) - 0.5f;
    const V3 point = (((k.aspect * (n_u * right)) - (n_v * up)) + pos) + mk(k.fd[0], k.fd[1], k.fd[2]);
    V3 ray_dir = normalized(point - pos);
    V3 ray_orig = pos;
    if (cam.aperture > 0.01f) {
        const float r = __builtin_sqrtf(xi[2]) * k.half_aperture;
        const float dx = cos_rev(xi[3]) * r, dy = sin_rev(xi[3]) * r;
        const V3 off = (dx * right) - (dy * up);
        const V3 focal_point = ray_orig + (cam.focus_distance * (1.0f / dot3(dir, ray_dir))) * ray_dir;
        ray_orig = ray_orig + off;
        ray_dir = normalized(focal_point - ray_orig);
    }
    ro = ray_orig; rd = ray_dir;
}

/* One bounce of color() (src/renderer.cu:106-138) after a hit: samples the material, plays the
 * specular/diffuse roulette and replaces the ray.  xi = Philox block 1+bounce. */
template <bool STATS>
__device__ __forceinline__ void shade_hit(const PtSceneDev &sc, const Best &best, const Surface &sf, const float xi[4],
                                          V3 &o, V3 &d, V3 &atten, V3 &result, Stats &st) {
    PT_SHADING_FP
    const PtMaterial &m = sc.materials[sc.entities[best.entity].material];
    MatSample ms;
    sample_material<STATS>(m, sf.u, sf.v, ms, st);
    const V3 ts_normal = ms.ts_normal, specular = ms.specular;
    V3 albedo = ms.albedo;
    const float smoothness = ms.smoothness;
    const float emission = m.emission;
    const V3 normal = to_world(ts_normal, tangent_space(sf.normal));

    albedo = mk(fminf(1.0f - specular.x, albedo.x), fminf(1.0f - specular.y, albedo.y), fminf(1.0f - specular.z, albedo.z));
    float spec_chance = luminosity(specular);
    float diff_chance = luminosity(albedo);
    const float sum = spec_chance + diff_chance;
    const float rsum = fast_rcp(sum);
    spec_chance *= rsum;
    diff_chance *= rsum;
    const float roulette = xi[0];
    o = sf.position + 0.001f * normal;
    /* the two outcomes of the roulette (src/renderer.cu:117-137) differ in the lobe that is sampled -- around the mirror
     * direction with exponent 1000^(s^2), or around the normal with exponent 1 -- and in the weight: the lobe's axis and
     * exponent are selected first, so that the sampling code runs once for the lanes of both outcomes */
    const bool spec = roulette < spec_chance;
    const float dn = dot3(d, normal);
    const V3 refl = d - (2.0f * dn) * normal;                                                       /* renderer.cu:147-149 */
    const V3 axis = spec ? refl : normal;
    const float alpha = spec ? __builtin_amdgcn_exp2f(smoothness * smoothness * 9.965784284662087f) : 1.0f;   /* 1000^(s^2) */
    if (!spec) result = result + emission * (albedo * atten);
    d = sample_hemisphere(axis, alpha, xi[1], xi[2]);
    const V3 tint = spec ? specular : albedo;
    const float k = spec ? fast_rcp(spec_chance) * ((alpha + 2.0f) * fast_rcp(alpha + 1.0f)) * dot3(normal, d) : fast_rcp(diff_chance);
    atten = mk(atten.x * tint.x * k, atten.y * tint.y * k, atten.z * tint.z * k);
}

/* counters: reduce over the wave, one atomic per wave and counter */
__device__ __forceinline__ void flush_stat(unsigned long long *dst, uint32_t v) {
    unsigned long long x = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, x);
}

/* least number of finished lanes (M_DONE) for which the wave leaves the geometry blocks to shade / refill them */
#ifndef PT_DONE_QUOTA
#define PT_DONE_QUOTA 12
#endif
static constexpr int kDoneQuota = PT_DONE_QUOTA;
/* triangle runs waiting in a wave at which it tests them (cooperative leaf phase) before walking on */
#ifndef PT_LEAF_QUOTA
#define PT_LEAF_QUOTA 20
#endif
static constexpr int kLeafQuota = PT_LEAF_QUOTA;
/* Waves per SIMD (= resident workgroups per CU) the two builds are compiled for: 4, i.e. 128 registers, with which the
 * lane state stays in registers through the shading code.  At 5 (96 registers) some 40 words per lane were spilled
 * and reloaded around every shade / refill round -- over a terabyte of scratch writes per 2048-spp frame -- for no gain
 * in speed: the kernel is bound by vector issue, which four waves saturate.  Deep octrees leave room for fewer
 * workgroups (LDS stacks): the heavy build is also instantiated for 3. */
#ifndef PT_WAVES_HEAVY
#define PT_WAVES_HEAVY 4
#endif
#ifndef PT_WAVES_LIGHT
#define PT_WAVES_LIGHT 4
#endif
static constexpr int kWavesHeavy = PT_WAVES_HEAVY, kWavesLight = PT_WAVES_LIGHT;

/* ------------------------------------------------------------------ closest-hit query kernel */
template <bool LOG>
__global__ void __launch_bounds__(kBlock) trace_kernel(PtSceneDev sc, const float *__restrict__ rays, int64_t n,
                                                       pt_hit *__restrict__ hits, int32_t *__restrict__ log, int log_stride,
                                                       unsigned long long *__restrict__ stats) {
    extern __shared__ float4 lds_raw[];
    float4 *lds_stack = lds_raw + kConstQuads;
    const bool lds_ents = load_lds_constants(lds_raw, sc);
    float4 *stack = lds_stack + threadIdx.x;
    uint32_t *stack1 = reinterpret_cast<uint32_t *>(lds_stack + stack_levels(sc.max_depth) * kBlock) + threadIdx.x;
    CoopLds *coop = coop_area(lds_stack, sc.max_depth);
    Stats st;
    stats_zero(st);
    Lane L;
    TriLog lg;
    lg.p = nullptr; lg.cap = log_stride;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x - step;
    L.mode = M_DONE;            /* "finished ray -1": the first trip only fetches */
    L.tri_left = 0u;
    bool first = true;
    for (;;) {
        if (L.mode == M_DONE) {
            if (!first) {
                pt_hit out;
                out.entity = -1; out.triangle = -1; out.t = L.best.t;
                out.bu = out.bv = out.u = out.v = 0.f;
                out.position[0] = out.position[1] = out.position[2] = 0.f;
                out.normal[0] = out.normal[1] = out.normal[2] = 0.f;
                if (L.best.entity >= 0) {
                    Surface sf;
                    finish_hit(sc, L.o, L.d, L.best, sf);
                    out.entity = L.best.entity; out.triangle = best_triangle(sc, L.best);
                    out.bu = L.best.bu; out.bv = L.best.bv; out.u = sf.u; out.v = sf.v;
                    out.position[0] = sf.position.x; out.position[1] = sf.position.y; out.position[2] = sf.position.z;
                    out.normal[0] = sf.normal.x; out.normal[1] = sf.normal.y; out.normal[2] = sf.normal.z;
                }
                hits[i] = out;
            }
            i += step;
            if (i < n) {
                st.segments++;
                lane_start_ray<keeps_fracs(!LOG, LOG, false)>(L, mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]),
                               mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
                if (LOG) entity_prepass<true, keeps_fracs(!LOG, LOG, false)>(sc, L, st);          /* (the logging build is the one without the cooperative phase) */
                if (LOG) { lg.p = log + i * log_stride; lg.p[0] = 0; }
            } else {
                L.mode = M_IDLE;
            }
        }
        first = false;
        if (__ballot(L.mode != M_IDLE) == 0ull) break;
        trace_wave<true, LOG, !LOG, LOG, false>(sc, L, st, lg, stack, stack1, kBlock, kDoneQuota, kLeafQuota, coop, threadIdx.x & 63, lds_ents);
    }
    if (stats) {
        flush_stat(&stats[0], st.segments);
        flush_stat(&stats[1], st.aabb_tests);
        flush_stat(&stats[2], st.node_visits);
        flush_stat(&stats[3], st.tri_tests);
    }
}

__global__ void __launch_bounds__(kBlock) camera_rays_kernel(pt_camera cam, CamConsts camk, uint32_t seed, int sample, float *__restrict__ rays) {
    const int n = cam.res_x * cam.res_y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float xi[4];
    rng_block((uint32_t)i, (uint32_t)sample, 0u, seed, xi);
    V3 o, d;
    camera_ray(cam, camk, i % cam.res_x, i / cam.res_x, xi, o, d);
    float *r = rays + 6 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}

/* ------------------------------------------------------------------ parity queries of the shading inputs */
/* the environment lookup of the render kernel for caller-supplied directions */
__global__ void __launch_bounds__(kBlock) sample_hdri_kernel(PtSceneDev sc, const float *__restrict__ dirs, int64_t n,
                                                             float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Stats st;
    const V3 c = sample_hdri<false>(sc, mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), st);
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}

/* the four material samples at the head of shade_hit for caller-supplied (u, v): out = n x 10 floats
 * {albedo rgb, specular rgb, smoothness, tangent-space normal xyz} */
__global__ void __launch_bounds__(kBlock) sample_material_kernel(PtSceneDev sc, int material, const float *__restrict__ uv,
                                                                 int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const PtMaterial &m = sc.materials[material];
    const float u = uv[2 * i], v = uv[2 * i + 1];
    MatSample ms;
    Stats st;
    sample_material<false>(m, u, v, ms, st);
    const V3 albedo = ms.albedo, specular = ms.specular, ts_normal = ms.ts_normal;
    const float smoothness = ms.smoothness;
    float *o = out + 10 * i;
    o[0] = albedo.x; o[1] = albedo.y; o[2] = albedo.z;
    o[3] = specular.x; o[4] = specular.y; o[5] = specular.z;
    o[6] = smoothness;
    o[7] = ts_normal.x; o[8] = ts_normal.y; o[9] = ts_normal.z;
}

/* ------------------------------------------------------------------ scene arrays laid out for their access (run once, at commit) */
/* texel records of a material (PtMaterial::packed) from its separate row-major maps; one thread per record, padding included */
__global__ void __launch_bounds__(kBlock) pack_maps_kernel(const uint8_t *__restrict__ albedo, const uint8_t *__restrict__ specular,
                                                           const uint8_t *__restrict__ normal, const uint8_t *__restrict__ smooth, int w, int h,
                                                           uint32_t tiles_x, uint32_t n_records, uint32_t shift, uint8_t *__restrict__ out) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_records) return;
    const uint32_t tile = r >> 4, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = (int)(tx * 4u + (r & 3u)), y = (int)(ty * 4u + ((r >> 2) & 3u));
    uint32_t w0 = 0u, w1 = 0u, w2 = 0u;
    if (x < w && y < h) {
        const size_t t = (size_t)y * (size_t)w + (size_t)x;
        if (albedo) w0 |= (uint32_t)albedo[3 * t] | ((uint32_t)albedo[3 * t + 1] << 8) | ((uint32_t)albedo[3 * t + 2] << 16);
        if (normal) {
            w0 |= (uint32_t)normal[3 * t] << 24;
            w1 |= (uint32_t)normal[3 * t + 1] | ((uint32_t)normal[3 * t + 2] << 8);
        }
        if (smooth) w1 |= (uint32_t)smooth[t] << 16;
        if (specular) w2 |= (uint32_t)specular[3 * t] | ((uint32_t)specular[3 * t + 1] << 8) | ((uint32_t)specular[3 * t + 2] << 16);
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(out + ((size_t)r << shift));
    o[0] = w0; o[1] = w1;
    if (shift == 4u) { o[2] = w2; o[3] = 0u; }
}

/* the records of bilinear environment look-ups (PtSceneDev::hdri_quads) from the row-major image */
__global__ void __launch_bounds__(kBlock) hdri_quads_kernel(const float *__restrict__ img, int w, int h, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint32_t)w * (uint32_t)h) return;
    const uint32_t y = i / (uint32_t)w, x = i - y * (uint32_t)w;
    const uint32_t xr = x + 1u == (uint32_t)w ? 0u : x + 1u, yr = y + 1u == (uint32_t)h ? 0u : y + 1u;     /* (x + 1) % w, (y + 1) % h: src/scene.cuh:62-65 */
    const float *a = img + 3 * ((size_t)y * w + x), *b = img + 3 * ((size_t)y * w + xr);
    const float *c = img + 3 * ((size_t)yr * w + x), *d = img + 3 * ((size_t)yr * w + xr);
    float4 *o = reinterpret_cast<float4 *>(out + 16 * (size_t)i);
    o[0] = make_float4(a[0], a[1], a[2], b[0]);
    o[1] = make_float4(b[1], b[2], c[0], c[1]);
    o[2] = make_float4(c[2], d[0], d[1], d[2]);
    o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_pack_maps(const uint8_t *const d_maps[4], int w, int h, uint32_t tiles_x, uint32_t n_records, uint32_t shift, uint8_t *d_out, hipStream_t stream) {
    hipLaunchKernelGGL(pack_maps_kernel, dim3((n_records + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_maps[PT_MAP_ALBEDO], d_maps[PT_MAP_SPECULAR],
                       d_maps[PT_MAP_NORMAL], d_maps[PT_MAP_SMOOTHNESS], w, h, tiles_x, n_records, shift, d_out);
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_hdri_quads(const float *d_img, int w, int h, float *d_out, hipStream_t stream) {
    const uint32_t n = (uint32_t)w * (uint32_t)h;
    hipLaunchKernelGGL(hdri_quads_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_img, w, h, d_out);
    PT_HIP(hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ persistent render kernel */
void slice_range(const SlicePlan &pl, int slice, int &s0, int &s1) {
    const int z = slice / pl.slices_per_split, j = slice - z * pl.slices_per_split;
    const long long N = pl.n_samples;
    const int zb = pl.sample_begin + (int)((N * z) / pl.n_split);
    const int ze = pl.sample_begin + (int)((N * (z + 1)) / pl.n_split);
    const long long M = ze - zb;
    s0 = zb + (int)((M * j) / pl.slices_per_split);
    s1 = zb + (int)((M * (j + 1)) / pl.slices_per_split);
}

__device__ __forceinline__ uint32_t udiv(uint32_t n, const UDiv &d) {
    const uint32_t t = __umulhi(d.mul, n);
    return (t + ((n - t) >> d.sh1)) >> d.sh2;
}

template <bool STATS, int WAVES, bool COOP, bool PREPASS, bool MULTI>
__global__ void __launch_bounds__(kBlock, WAVES) render_kernel(RenderParams P) {
    static_assert(!(PREPASS && MULTI), "the entity pre-pass and the weighted entity rule are alternatives");
    extern __shared__ float4 lds_raw[];
    float4 *lds_stack = lds_raw + kConstQuads;
    const bool lds_ents = load_lds_constants(lds_raw, P.sc);
    float4 *stack = lds_stack + threadIdx.x;
    uint32_t *stack1 = reinterpret_cast<uint32_t *>(lds_stack + stack_levels(P.sc.max_depth) * kBlock) + threadIdx.x;
    CoopLds *coop = coop_area(lds_stack, P.sc.max_depth);
    const int lane = threadIdx.x & 63;
    const int W = P.cam.res_x, H = P.cam.res_y;
    const int n_slices = P.n_slices;
    const size_t n_pixels = (size_t)W * (size_t)H;

    /* wave-uniform pool: the item this wave is handing out pixels of, decoded once when it is pulled (on the scalar
     * unit: every value below is the same in all lanes): first pixel of its 8x8 tile, slice, the slice's samples */
    int pool_px0 = 0, pool_py0 = 0, pool_slice = 0, pool_s0 = 0, pool_s1 = 0;
    int pool_next = 64;
    bool exhausted = false;
    /* Work queues.  The item space (macro-tile order, all slices of a tile adjacent) is cut into one
     * contiguous range per workgroup, so the waves of a workgroup -- which share a CU and its L1 --
     * trace neighbouring pixels; workgroups b, b+8, ... tend to share an XCD and its L2 (a speed
     * matter only), so they get adjacent ranges.  A wave whose range is empty steals from the next. */
    const uint32_t nq = P.n_queues, per_queue = P.items_per_queue;
    uint32_t q;
    {
        const uint32_t b = blockIdx.x, per = nq / 8u, rem = nq % 8u, x = b % 8u;
        q = (x < rem ? x * (per + 1u) : rem * (per + 1u) + (x - rem) * per) + b / 8u;      /* bijective on [0, nq) */
    }
    uint32_t probes = 0;

    /* lane state: the (pixel, slice) it owns, the path it is extending, the ray it is tracing.  The small fields share
     * one register (they are live across the octree walk, where registers are short):
     *   meta bits 0..15  samples of the slice still to do, the current one included (0: the lane owns no pixel)
     *        bits 16..21 slice;  bits 22..24 bounce;  bit 25 no more work for this lane;  bit 26 the lane owns a pixel whose
     *        sum has not been written yet */
    constexpr uint32_t kLeftMask = 0xFFFFu, kSliceShift = 16u, kBounceShift = 22u, kBounceOne = 1u << kBounceShift,
                       kBounceMask = 7u << kBounceShift, kDoneBit = 1u << 25, kOwnsBit = 1u << 26;
    uint32_t meta = 0, pxy = 0;
    int s_cur = 0;
    V3 accum = mk(0.f, 0.f, 0.f), result = mk(0.f, 0.f, 0.f), atten = mk(1.f, 1.f, 1.f);
    Lane L;
    L.mode = M_IDLE;
    L.tri_left = 0u;
    Stats st;
    stats_zero(st);
    const uint64_t cyc_begin = STATS ? wave_clock() : 0;
    const uint64_t real_begin = STATS ? __builtin_amdgcn_s_memrealtime() : 0;      /* (the constant 100 MHz clock is the one all XCDs share) */
    TriLog lg;
    lg.p = nullptr; lg.cap = 0;

    for (;;) {
        /* ---- shade the lanes whose ray is finished (color()'s loop body, src/renderer.cu:99-139) */
        if (STATS) st.front_trips++;
        /* a finished ray that hit nothing takes the environment and ends its path here; one that hit something is
         * shaded further down, after the refill, so that the random numbers of both kinds of lane -- the shading block of
         * a hit, the camera block of a new path -- come from one evaluation of the generator */
        const bool hit = L.mode == M_DONE && L.best.entity >= 0;
        if (L.mode == M_DONE) {
            if (STATS) st.front_lanes++;
            if (!hit) {
                const V3 env = sample_hdri<STATS>(P.sc, L.d, st);
                result = result + atten * env;
                accum = accum + result;
                s_cur++;
                meta -= 1u;
                L.mode = M_IDLE;
            }
        }
        /* ---- lanes without a path: finish their (pixel, slice), take a new one, start the next sample */
        /* (a lane that owns a pixel and is idle has just ended a path) */
        const bool wants = L.mode == M_IDLE && (meta & (kLeftMask | kDoneBit)) == 0u;
        if (wants && (meta & kOwnsBit)) {
            const size_t pixel = (size_t)(pxy >> 16) * (size_t)W + (size_t)(pxy & 0xFFFFu);
            float *out = P.partials + 3 * ((size_t)((meta >> kSliceShift) & 63u) * n_pixels + pixel);
            out[0] = accum.x; out[1] = accum.y; out[2] = accum.z;
            meta &= ~kOwnsBit;
        }
        const unsigned long long need = __ballot(wants);
        if (need != 0ull) {
            const int n_need = __popcll(need);
            const int rank = __popcll(need & ((1ull << lane) - 1ull));
            int assigned = 0;
            bool got = false;
            int my_slot = 0, my_px0 = 0, my_py0 = 0, my_slice = 0, my_s0 = 0, my_s1 = 0;
            while (assigned < n_need) {
                if (pool_next >= 64) {
                    if (exhausted) break;
                    uint32_t id = 0xFFFFFFFFu;
                    if (COOP) {
                        /* The builds with the cooperative phase try the ranges one after the other (their lanes have no register to
                         * spare: the map search below cost them three more spilled registers and 1 % of every launch, and their
                         * launches, scenes of long walks, lose little to the end of a launch) */
                        if (lane == 0) {
                            for (;;) {
                                const unsigned long long first = (unsigned long long)per_queue * q;
                                const uint32_t lo = (uint32_t)(first < P.n_items ? first : P.n_items);
                                const uint32_t hi = (uint32_t)(first + per_queue < P.n_items ? first + per_queue : P.n_items);
                                const uint32_t t = atomicAdd(&P.tickets[q], 1u);
                                if (t < hi - lo) { id = lo + t; break; }
                                /* this range is finished for good; the first draw past its end reports it, and once
                                 * every range is reported nothing is left anywhere */
                                if (t == hi - lo) atomicAdd(&P.tickets[nq], 1u);
                                if (__hip_atomic_load(&P.tickets[nq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nq) break;
                                q = (q + 1u == nq) ? 0u : q + 1u;
                                if (++probes >= nq) break;               /* every range has been seen empty */
                            }
                        }
                        id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
                        q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
                        probes = (uint32_t)__builtin_amdgcn_readfirstlane((int)probes);
                    } else {
                        for (;;) {
                            /* queue q holds items [q * per, min((q + 1) * per, n_items)); one lane draws */
                            uint32_t past = 0u;               /* 1: the draw was the first one past the range's end */
                            if (lane == 0) {
                                const unsigned long long first = (unsigned long long)per_queue * q;
                                const uint32_t lo = (uint32_t)(first < P.n_items ? first : P.n_items);
                                const uint32_t hi = (uint32_t)(first + per_queue < P.n_items ? first + per_queue : P.n_items);
                                const uint32_t t = atomicAdd(&P.tickets[q], 1u);
                                if (t < hi - lo) id = lo + t;
                                else if (t == hi - lo) {
                                    /* This range is finished for good: the first draw past its end sets the range's bit in the map
                                     * of exhausted ranges. */
                                    atomicOr(&P.tickets[kTicketMapWord + (q >> 5)], 1u << (q & 31u));
                                    past = 1u;
                                }
                            }
                            id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
                            if (id != 0xFFFFFFFFu) break;
                            past = (uint32_t)__builtin_amdgcn_readfirstlane((int)past);
                            /* A wave without work reads the map (one word per lane) and goes straight to the next range after its
                             * own that is not marked -- trying the ranges one by one, an atomic round trip each, cost a wave of an
                             * unevenly lit image hundreds of trips at the end of a launch (teapot scene: 1.7 ms of a 27 ms launch). */
                            const uint32_t n_map = (nq + 31u) >> 5;                        /* at most 64 words: kMaxQueues */
                            uint32_t w = ~0u;
                            if ((uint32_t)lane < n_map) {
                                w = __hip_atomic_load(P.tickets + kTicketMapWord + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const uint32_t bits_here = nq - 32u * (uint32_t)lane;          /* ranges this word stands for */
                                if (bits_here < 32u) w |= ~0u << bits_here;
                            }
                            if (past != 0u && (uint32_t)lane == (q >> 5)) w |= 1u << (q & 31u);        /* (this wave's own mark may not be visible yet) */
                            const uint32_t open_ranges = ~w;
                            const uint32_t start = q + 1u;                                 /* first candidate, cyclically */
                            const uint32_t ls = (start >> 5) & 63u;
                            const uint32_t from = ~0u << (start & 31u);
                            const bool wraps = start >= nq;
                            const uint32_t after = wraps ? 0u : ((uint32_t)lane == ls ? open_ranges & from : ((uint32_t)lane > ls ? open_ranges : 0u));
                            const uint32_t before = wraps ? open_ranges : ((uint32_t)lane == ls ? open_ranges & ~from : ((uint32_t)lane < ls ? open_ranges : 0u));
                            unsigned long long lanes_with = __ballot(after != 0u);
                            uint32_t pick = after;
                            if (lanes_with == 0ull) { lanes_with = __ballot(before != 0u); pick = before; }
                            if (lanes_with == 0ull) break;                                 /* every range is marked: nothing is left anywhere */
                            const int src = __builtin_ctzll(lanes_with);
                            q = 32u * (uint32_t)src + (uint32_t)__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)pick, src));
                        }
                    }
                    if (id == 0xFFFFFFFFu) { exhausted = true; break; }
                    /* item -> (tile in macro-tile order, slice) */
                    const uint32_t tidx = udiv(id, P.div_slices);
                    pool_slice = (int)(id - tidx * (uint32_t)n_slices);
                    const uint32_t tseq = tidx * P.tile_stride + P.tile_first;      /* this launch's share of the tiles, interleaved */
                    const uint32_t mt = tseq >> 6, lt = tseq & 63u;
                    const uint32_t mty = udiv(mt, P.div_mtiles_x), mtx = mt - mty * (uint32_t)P.mtiles_x;
                    pool_px0 = (int)((mtx * 8u + (lt & 7u)) * 8u);
                    pool_py0 = (int)((mty * 8u + (lt >> 3)) * 8u);
                    pool_s0 = P.slice_s0[pool_slice];
                    pool_s1 = P.slice_s1[pool_slice];
                    pool_next = 0;
                }
                const int k = min(n_need - assigned, 64 - pool_next);
                if (wants && rank >= assigned && rank < assigned + k) {
                    got = true;
                    my_slot = pool_next + (rank - assigned);
                    my_px0 = pool_px0; my_py0 = pool_py0; my_slice = pool_slice; my_s0 = pool_s0; my_s1 = pool_s1;
                }
                pool_next += k;
                assigned += k;
            }
            if (wants) {
                if (!got) {
                    meta = kDoneBit;
                } else {
                    const int px = my_px0 + (my_slot & 7), py = my_py0 + (my_slot >> 3);
                    if (px < W && py < H) {
                        pxy = ((uint32_t)py << 16) | (uint32_t)px;
                        s_cur = my_s0;
                        meta = kOwnsBit | ((uint32_t)my_slice << kSliceShift) | (uint32_t)(my_s1 - my_s0);
                        accum = mk(0.f, 0.f, 0.f);
                    }
                }
            }
        }
        const bool start = L.mode == M_IDLE && (meta & kLeftMask) != 0u;
        if (hit || start) {
            float xi[4];
            const uint32_t bounce = (meta & kBounceMask) >> kBounceShift;
            rng_block((pxy >> 16) * (uint32_t)W + (pxy & 0xFFFFu), (uint32_t)s_cur, hit ? 1u + bounce : 0u, P.seed, xi);
            /* both kinds of lane end with a ray to trace (but for a path at the bounce limit): the ray is started once,
             * after the two branches, so that its set-up is not there twice with copies where the branches join */
            V3 o = L.o, d = L.d;
            bool go = true;
            if (hit) {
                Surface sf;
                finish_hit(P.sc, L.o, L.d, L.best, sf);
                shade_hit<STATS>(P.sc, L.best, sf, xi, o, d, atten, result, st);
                meta += kBounceOne;
                if (bounce + 1u >= 6u) {   /* src/renderer.cu:99; the lane's next sample starts in the next round */
                    accum = accum + result;
                    s_cur++;
                    meta -= 1u;
                    L.mode = M_IDLE;
                    go = false;
                }
            } else {
                camera_ray(P.cam, P.camk, (int)(pxy & 0xFFFFu), (int)(pxy >> 16), xi, o, d);
                atten = mk(1.f, 1.f, 1.f);
                result = mk(0.f, 0.f, 0.f);
                meta &= ~kBounceMask;
                if (STATS) st.paths++;
            }
            if (go) {
                if (STATS) st.segments++;
                lane_start_ray<keeps_fracs(COOP, PREPASS, MULTI)>(L, o, d);
                if (PREPASS) entity_prepass<STATS, keeps_fracs(COOP, PREPASS, MULTI)>(P.sc, L, st);
            }
        }
        if (__ballot(L.mode != M_IDLE) == 0ull) {
            if (__ballot(!(meta & kDoneBit)) == 0ull) break;      /* every lane is out of work: the grid drains */
            continue;                                /* e.g. lanes that drew an out-of-image slot ask again */
        }
        const uint64_t c0 = STATS ? wave_clock() : 0;
        trace_wave<STATS, false, COOP, PREPASS, MULTI>(P.sc, L, st, lg, stack, stack1, kBlock, P.done_quota, P.leaf_quota, coop, lane, lds_ents);
        if (STATS) st.cyc_trace += wave_clock() - c0;
    }
    if (STATS && P.stats) {
        flush_stat(&P.stats[0], st.segments);
        flush_stat(&P.stats[1], st.aabb_tests);
        flush_stat(&P.stats[2], st.node_visits);
        flush_stat(&P.stats[3], st.tri_tests);
        flush_stat(&P.stats[4], st.env_lookups);
        flush_stat(&P.stats[5], st.tex_fetches);
        flush_stat(&P.stats[6], st.paths);
        flush_stat(&P.stats[7], st.ent_lanes);
        flush_stat(&P.stats[8], st.node_lanes);
        flush_stat(&P.stats[9], st.leaf_lanes);
        flush_stat(&P.stats[10], st.front_lanes);
        /* trips are wave-uniform: count them once per wave */
        flush_stat(&P.stats[11], lane == 0 ? st.ent_trips : 0u);
        flush_stat(&P.stats[12], lane == 0 ? st.node_trips : 0u);
        flush_stat(&P.stats[13], lane == 0 ? st.leaf_trips : 0u);
        flush_stat(&P.stats[14], lane == 0 ? st.front_trips : 0u);
        flush_stat(&P.stats[20], st.idle_done);
        flush_stat(&P.stats[21], st.idle_leaf);
        flush_stat(&P.stats[22], st.idle_other);
        /* wave life times: in the blocks, in trace_wave, and from the kernel's first instruction to here */
        if (lane == 0) {
            atomicAdd(&P.stats[15], (unsigned long long)st.cyc_ent);
            atomicAdd(&P.stats[16], (unsigned long long)st.cyc_node);
            atomicAdd(&P.stats[17], (unsigned long long)st.cyc_leaf);
            atomicAdd(&P.stats[18], (unsigned long long)st.cyc_trace);
            const uint64_t cyc_end = wave_clock();
            atomicAdd(&P.stats[19], (unsigned long long)(cyc_end - cyc_begin));
            /* when the waves started and ended, in ticks of the 100 MHz clock: sum and earliest start, latest end, number of waves */
            const uint64_t real_end = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&P.stats[23], (unsigned long long)real_begin);
            atomicMax(&P.stats[24], ~(unsigned long long)real_begin);
            atomicMax(&P.stats[25], (unsigned long long)real_end);
            atomicAdd(&P.stats[26], 1ull);
            atomicAdd(&P.stats[27], (unsigned long long)(real_end - real_begin));
        }
    }
}

/* split[z][p] += sum_j partial[z * slices_per_split + j][p], j ascending */
__global__ void __launch_bounds__(kBlock) compound_partials_kernel(const float *__restrict__ partials, float *__restrict__ split,
                                                                   int64_t n_floats, int n_split, int slices_per_split) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_floats) return;
    for (int z = 0; z < n_split; z++) {
        float s = 0.0f;
        for (int j = 0; j < slices_per_split; j++) s = s + partials[(int64_t)(z * slices_per_split + j) * n_floats + i];
        split[(int64_t)z * n_floats + i] = split[(int64_t)z * n_floats + i] + s;
    }
}

/* sum = (first ? 0 : sum) + part[0] + part[1] + ...; split += sum when the split buffer's last slice is among them: the
 * additions of compound_partials_kernel for one split buffer, in the same order, a few slices at a time */
__global__ void __launch_bounds__(kBlock) fold_partials_kernel(float *__restrict__ sum, const float *__restrict__ parts, int n_parts, int first,
                                                               float *__restrict__ split, int64_t n_floats) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_floats) return;
    float s = first ? 0.0f : sum[i];
    for (int j = 0; j < n_parts; j++) s = s + parts[(int64_t)j * n_floats + i];
    if (split) split[i] = split[i] + s;
    else sum[i] = s;
}

/* compound_buffers (src/post_processing.cu:243-256) */
__global__ void __launch_bounds__(kBlock) compound_split_kernel(const float *__restrict__ split, float *__restrict__ out,
                                                                int64_t n_floats, int n_split) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_floats) return;
    float s = 0.0f;
    for (int z = 0; z < n_split; z++) s = s + split[(int64_t)z * n_floats + i];
    out[i] = s;
}

/* ------------------------------------------------------------------ launchers */
SlicePlan make_slice_plan(int sample_begin, int sample_end, int n_split, int64_t n_tiles) {
    SlicePlan p;
    p.sample_begin = sample_begin;
    p.n_samples = sample_end - sample_begin;
    p.n_split = n_split;
    const int per_split = (p.n_samples + n_split - 1) / n_split;
    /* slices of about 8 samples, at most 64 slices over all split buffers: a work item is one 8x8 tile
     * of one slice, and the end of a launch is as ragged as its largest items are long (a 256-spp frame
     * renders 6 % faster with slices of 8 than of 32; beyond 16 slices per buffer the tail is under 1 %
     * and the workspace, one image per slice, stops growing; slices of 4 cost a launch with plenty of items 6 %).
     * A launch so small that slices of 8 leave it under 32 k items -- eight per wave of a full grid -- is cut into
     * slices of 4 (cornell box 640x360, 32 spp: 1.18 -> 1.07 ms; shorter ones bring nothing more). */
    int cap = 64 / n_split;
    if (cap < 1) cap = 1;
    static const int forced_len = [] { const char *e = getenv("PTAMD_SLICE_SAMPLES"); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 0; }();
    int len = 8;
    if (n_tiles * n_split * (int64_t)std::min(cap, std::max(1, (per_split + 7) / 8)) < 32768) len = 4;
    if (forced_len) len = forced_len;
    int s = (per_split + len - 1) / len;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    p.slices_per_split = s;
    return p;
}

static size_t stack_bytes(int max_depth) {
    return (size_t)kConstQuads * 16 + (size_t)stack_levels(max_depth) * kBlock * (sizeof(float4) + sizeof(uint32_t)) + (kBlock / 64) * sizeof(CoopLds);
}

int launch_trace(const PtSceneDev &sc, const float *d_rays, int64_t n, pt_hit *d_hits, int32_t *d_log, int log_stride,
                 unsigned long long *d_stats, hipStream_t stream) {
    if (n <= 0) return 0;
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = stack_bytes(sc.max_depth);
    if (d_log)
        hipLaunchKernelGGL(trace_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), lds, stream, sc, d_rays, n, d_hits, d_log, log_stride, d_stats);
    else
        hipLaunchKernelGGL(trace_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), lds, stream, sc, d_rays, n, d_hits, d_log, log_stride, d_stats);
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_sample_hdri(const PtSceneDev &sc, const float *d_dirs, int64_t n, float *d_out, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sample_hdri_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sc, d_dirs, n, d_out);
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_sample_material(const PtSceneDev &sc, int material, const float *d_uv, int64_t n, float *d_out, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sample_material_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sc, material, d_uv, n, d_out);
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_camera_rays(const pt_camera &cam, uint32_t seed, int sample, float *d_rays, hipStream_t stream) {
    const int n = cam.res_x * cam.res_y;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(camera_rays_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cam, make_cam_consts(cam), seed, sample, d_rays);
    PT_HIP(hipGetLastError());
    return 0;
}

/* workgroups of the render kernel that stay resident per CU: bounded by registers (`waves` waves per SIMD = as
 * many workgroups of 256 threads per CU) and by LDS (160 KiB / the stacks of one workgroup) */
int render_workgroups_per_cu(int max_depth, bool heavy) {
    const size_t lds = stack_bytes(max_depth);
    int per_cu = heavy ? kWavesHeavy : kWavesLight;
    const int by_lds = (int)((160 * 1024) / (lds ? lds : 1));
    if (by_lds < per_cu) per_cu = by_lds > 0 ? by_lds : 1;
    return per_cu;
}

int launch_render(RenderParams p, int n_cu, int max_queues, bool heavy, hipStream_t stream) {
    const size_t lds = stack_bytes(p.sc.max_depth);
    /* persistent grid: as many workgroups as stay resident */
    const int per_cu = render_workgroups_per_cu(p.sc.max_depth, heavy);
    int64_t blocks = (int64_t)n_cu * per_cu;
    const int64_t max_useful = ((int64_t)p.n_items * 64 + kBlock - 1) / kBlock;
    if (blocks > max_useful) blocks = max_useful;
    if (blocks < 1) blocks = 1;
    if (blocks > max_queues) blocks = max_queues;
    p.n_queues = (uint32_t)blocks;
    p.items_per_queue = (uint32_t)(((uint64_t)p.n_items + (uint64_t)blocks - 1) / (uint64_t)blocks);
    const dim3 grid((unsigned)blocks), block(kBlock);
    /* scenes of several entities without the entity pre-pass run the instantiations with the weighted entity rule (trace_wave) */
    const bool prepass = !heavy && p.sc.n_entities >= kPrepassMinEntities && p.sc.n_entities <= 31;
    const bool multi = !prepass && p.sc.n_entities > 1;
#define PT_LAUNCH(WAVES, COOP, PREPASS, MULTI)                                                                                  \
    do {                                                                                                                        \
        if (p.stats) hipLaunchKernelGGL((render_kernel<true, WAVES, COOP, PREPASS, MULTI>), grid, block, lds, stream, p);        \
        else hipLaunchKernelGGL((render_kernel<false, WAVES, COOP, PREPASS, MULTI>), grid, block, lds, stream, p);               \
    } while (0)
    if (heavy) {
        /* deep octrees need so much LDS for their stacks that fewer workgroups fit: the build for 3 waves per SIMD
         * then has the larger register budget for the same residency */
        if (per_cu >= kWavesHeavy) {
            if (multi) PT_LAUNCH(kWavesHeavy, true, false, true);
            else PT_LAUNCH(kWavesHeavy, true, false, false);
        } else {
            if (multi) PT_LAUNCH(3, true, false, true);
            else PT_LAUNCH(3, true, false, false);
        }
    } else {
        /* scenes of many small entities get the instantiation with the entity pre-pass */
        if (prepass) PT_LAUNCH(kWavesLight, false, true, false);
        else if (multi) PT_LAUNCH(kWavesLight, false, false, true);
        else PT_LAUNCH(kWavesLight, false, false, false);
    }
#undef PT_LAUNCH
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_compound_partials(const float *d_partials, float *d_split, int64_t n_pixels, const SlicePlan &plan, hipStream_t stream) {
    const int64_t n_floats = n_pixels * 3;
    hipLaunchKernelGGL(compound_partials_kernel, dim3((unsigned)((n_floats + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       d_partials, d_split, n_floats, plan.n_split, plan.slices_per_split);
    PT_HIP(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(kBlock) zero_words_kernel(uint32_t *__restrict__ p, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = 0u;
}

int launch_zero_words(uint32_t *d_words, int n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(zero_words_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_words, n);
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_fold_partials(float *d_sum, const float *d_parts, int n_parts, bool first, float *d_split, int64_t n_pixels, hipStream_t stream) {
    const int64_t n_floats = n_pixels * 3;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)((n_floats + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_sum, d_parts, n_parts,
                       first ? 1 : 0, d_split, n_floats);
    PT_HIP(hipGetLastError());
    return 0;
}

int launch_compound_split(const float *d_split, float *d_out, int64_t n_pixels, int n_split, hipStream_t stream) {
    const int64_t n_floats = n_pixels * 3;
    hipLaunchKernelGGL(compound_split_kernel, dim3((unsigned)((n_floats + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_split,
                       d_out, n_floats, n_split);
    PT_HIP(hipGetLastError());
    return 0;
}

} // namespace ptamd
