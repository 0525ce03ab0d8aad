/* TEST INFRASTRUCTURE — see cpu_ref.h.  Build with -O2 -ffp-contract=off (no FMA contraction,
 * no fast-math): every float operation below is then the same IEEE-754 binary32 operation, in
 * the same order, that the reference performs when its sources are built with the same flags.
 *
 * Argument-evaluation order: the reference draws several rand() values inside one constructor
 * call (material.h:46-48, main.cpp:57,63,68), whose order the language leaves unspecified.  The
 * pinned oracle build (g++ 11.4, x86-64) evaluates such arguments right to left; this file
 * writes that order out explicitly so that it does not depend on the C compiler used here.
 */
#include "cpu_ref.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <limits.h>

/* ------------------------------------------------------------------ vec3 (vec3.h:69-135, vec3.cpp:109-137) */
static orc_vec3 V(float x, float y, float z) { orc_vec3 v = { x, y, z }; return v; }
static orc_vec3 add(orc_vec3 a, orc_vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static orc_vec3 sub(orc_vec3 a, orc_vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static orc_vec3 mulv(orc_vec3 a, orc_vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static orc_vec3 smul(float k, orc_vec3 a) { return V(k * a.x, k * a.y, k * a.z); }
static orc_vec3 muls(orc_vec3 a, float k) { return V(a.x * k, a.y * k, a.z * k); }
static orc_vec3 divs(orc_vec3 a, float k) { return V(a.x / k, a.y / k, a.z / k); }
static orc_vec3 neg(orc_vec3 a) { return V(-a.x, -a.y, -a.z); }
static float dot(orc_vec3 a, orc_vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static orc_vec3 cross(orc_vec3 a, orc_vec3 b)
{
    return V(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
static float sqlen(orc_vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static float length(orc_vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static orc_vec3 unit(orc_vec3 a) { return divs(a, length(a)); }
static orc_vec3 point_at(orc_ray r, float t) { return add(r.o, smul(t, r.d)); }   /* ray.cpp:23-26 */

/* ------------------------------------------------------------------ Philox4x32-10 (Salmon et al., SC'11) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float orc_usrc_word(const orc_usrc *u, uint32_t group, int w)
{
    uint32_t ctr[4] = { u->px, u->smp, u->bounce, group }, key[2] = { (uint32_t)u->seed, (uint32_t)(u->seed >> 32) }, out[4];
    orc_philox4x32_10(ctr, key, out);
    return (float)(out[w] >> 8) * (1.0f / 16777216.0f);
}

void orc_usrc_slot(orc_usrc *u, uint32_t px, uint32_t smp, uint32_t bounce)
{
    u->px = px; u->smp = smp; u->bounce = bounce; u->draw = 0;
}

/* One uniform as the reference forms it: rand() / (RAND_MAX + 1.0), a double in [0,1). */
double orc_uniform(orc_usrc *u)
{
    int k;
    if (u->kind == 0) {
        if (u->pos >= u->n_replay) abort();
        k = u->replay[u->pos++];
    } else if (u->kind == 1) {
        k = (int)random();
    } else {
        uint32_t ctr[4] = { u->px, u->smp, u->bounce, u->draw / 4u }, key[2] = { (uint32_t)u->seed, (uint32_t)(u->seed >> 32) }, out[4];
        orc_philox4x32_10(ctr, key, out);
        k = (int)((out[u->draw % 4u] >> 8) << 7);
        u->draw += 1;
    }
    u->consumed += 1;
    return k / (RAND_MAX + 1.0);
}

/* ------------------------------------------------------------------ Sphere::Hit (sphere.cpp:15-52) */
int orc_sphere_hit(orc_ray r, orc_vec3 c, float radius, float tmin, float tmax, orc_hit *rec)
{
    orc_vec3 oc = sub(r.o, c);
    float a = dot(r.d, r.d);
    float b = dot(oc, r.d);
    float cc = dot(oc, oc) - radius * radius;
    float desc = b * b - a * cc;
    if (desc > 0.0f) {
        float temp = (-b - sqrtf(desc)) / a;
        if (temp < tmax && temp > tmin) {
            rec->t = temp; rec->p = point_at(r, temp); rec->n = divs(sub(rec->p, c), radius);
            return 1;
        }
        temp = (-b + sqrtf(desc)) / a;
        if (temp < tmax && temp > tmin) {
            rec->t = temp; rec->p = point_at(r, temp); rec->n = divs(sub(rec->p, c), radius);
            return 1;
        }
    }
    return 0;
}

/* HittableList::Hit (hittablelist.cpp:13-31): index order, shrinking tmax, strict '<'. */
int orc_world_hit(const orc_scene *s, orc_ray r, float tmin, float tmax, orc_hit *rec)
{
    orc_hit tmp; int any = 0; float closest = tmax;
    for (int i = 0; i < s->n; ++i) {
        if (orc_sphere_hit(r, V(s->cx[i], s->cy[i], s->cz[i]), s->radius[i], tmin, closest, &tmp)) {
            any = 1; closest = tmp.t; tmp.idx = i; *rec = tmp;
        }
    }
    return any;
}

/* ------------------------------------------------------------------ material.h helpers */
orc_vec3 orc_reflect(orc_vec3 v, orc_vec3 n) { return sub(v, smul(2 * dot(v, n), n)); }   /* :35-38 */

int orc_refract(orc_vec3 v, orc_vec3 n, float ni_over_nt, orc_vec3 *out)                  /* :16-33 */
{
    orc_vec3 uv = unit(v);
    float dt = dot(uv, n);
    float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (disc > 0) {
        *out = sub(smul(ni_over_nt, sub(uv, muls(n, dt))), muls(n, sqrtf(disc)));
        return 1;
    }
    return 0;
}

float orc_schlick(float cosine, float ref_idx)                                            /* :7-14 */
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    /* pow(float, int) promotes to double; the sum is formed in double and narrowed on return */
    return (float)((double)r0 + (double)(1.0f - r0) * pow((double)(1.0f - cosine), 5.0));
}

orc_vec3 orc_random_in_unit_sphere(orc_usrc *u)                                           /* :40-53 */
{
    orc_vec3 p;
    do {
        /* constructor arguments evaluated right to left by the pinned build: z, then y, then x */
        float z = (float)orc_uniform(u);
        float y = (float)orc_uniform(u);
        float x = (float)orc_uniform(u);
        p = sub(smul(2.0f, V(x, y, z)), V(1.0f, 1.0f, 1.0f));
    } while (sqlen(p) >= 1.0f);
    return p;
}

float orc_metal_fuzz(float fuzz) { return fuzz < 1 ? fuzz : 1.0f; }                        /* :73-83 */

int orc_scatter(int type, orc_vec3 albedo, float param, orc_ray in, const orc_hit *rec, orc_usrc *u, orc_vec3 *att, orc_ray *out)
{
    if (type == 0) {                                                                       /* Lambertian :107-116 */
        orc_vec3 target = add(add(rec->p, rec->n), orc_random_in_unit_sphere(u));
        out->o = rec->p; out->d = sub(target, rec->p);
        *att = albedo;
        return 1;
    }
    if (type == 1) {                                                                       /* Metal :85-91 */
        orc_vec3 reflected = orc_reflect(unit(in.d), rec->n);
        out->o = rec->p; out->d = add(reflected, smul(param, orc_random_in_unit_sphere(u)));
        *att = albedo;
        return dot(out->d, rec->n) > 0;
    }
    /* Dielectric :132-186 */
    orc_vec3 outward, refracted = V(0, 0, 0);
    orc_vec3 reflected = orc_reflect(in.d, rec->n);
    float ni_over_nt, reflect_prob, cosine;
    *att = V(1.0f, 1.0f, 1.0f);
    if (dot(in.d, rec->n) > 0) {
        outward = neg(rec->n);
        ni_over_nt = param;
        cosine = dot(in.d, rec->n) / length(in.d);
        cosine = sqrtf(1.0f - param * param * (1.0f - cosine * cosine));
    } else {
        outward = rec->n;
        ni_over_nt = 1.0f / param;
        cosine = -dot(in.d, rec->n) / length(in.d);
    }
    if (orc_refract(in.d, outward, ni_over_nt, &refracted)) reflect_prob = orc_schlick(cosine, param);
    else reflect_prob = 1.0f;
    out->o = rec->p;
    if (orc_uniform(u) < reflect_prob) out->d = reflected; else out->d = refracted;
    return 1;
}

/* ------------------------------------------------------------------ Camera (camera.cpp:6-29, :46-53) */
void orc_camera_init(orc_camera *c, orc_vec3 from, orc_vec3 at, orc_vec3 up, float vfov, float aspect, float aperture, float focus)
{
    c->lens_radius = aperture / 2;
    float theta = vfov * (float)M_PI / 180;
    float half_h = tanf(theta / 2);     /* <math.h> is in scope there, so tan(float) is the float overload */
    float half_w = aspect * half_h;
    c->origin = from;
    c->w = unit(sub(from, at));
    c->u = unit(cross(up, c->w));
    c->v = cross(c->w, c->u);
    c->llc = sub(sub(sub(c->origin, smul(half_w * focus, c->u)), smul(half_h * focus, c->v)), smul(focus, c->w));
    c->horiz = smul(2 * half_w * focus, c->u);
    c->vert = smul(2 * half_h * focus, c->v);
}

orc_ray orc_camera_get_ray(const orc_camera *c, float s, float t, float disk_x, float disk_y)
{
    orc_vec3 rd = smul(c->lens_radius, V(disk_x, disk_y, 0.0f));
    orc_vec3 offset = add(muls(c->u, rd.x), muls(c->v, rd.y));
    orc_ray r;
    r.o = add(c->origin, offset);
    r.d = sub(sub(add(add(c->llc, smul(s, c->horiz)), smul(t, c->vert)), c->origin), offset);
    return r;
}

/* ------------------------------------------------------------------ Color (main.cpp:13-39) */
static int g_max_depth = 50;     /* the literal 50 of main.cpp:22; settable so that the library's max_depth parameter can be checked */
void orc_set_max_depth(int d) { g_max_depth = d; }
/* RenderStats as main_modern.cpp:23-61 maintains it: Color() calls, calls whose Hit() was true, calls whose
 * Scatter() was true below the depth cap. */
static uint64_t g_counters[3];
void orc_counters_reset(void) { g_counters[0] = g_counters[1] = g_counters[2] = 0; }
void orc_counters_get(uint64_t out[3]) { out[0] = g_counters[0]; out[1] = g_counters[1]; out[2] = g_counters[2]; }
orc_vec3 orc_color(const orc_scene *s, orc_ray r, int depth, orc_usrc *u, uint64_t *rays)
{
    orc_hit rec;
    *rays += 1; g_counters[0] += 1;
    if (u->kind == 2) orc_usrc_slot(u, u->px, u->smp, u->bounce + 1);   /* query k of a sample scatters from slot k */
    if (orc_world_hit(s, r, 0.001f, FLT_MAX, &rec)) {
        orc_ray scattered; orc_vec3 att;
        int i = rec.idx;
        g_counters[1] += 1;
        if (depth < g_max_depth && orc_scatter(s->type[i], V(s->ar[i], s->ag[i], s->ab[i]), s->param[i], r, &rec, u, &att, &scattered)) {
            g_counters[2] += 1;
            return mulv(att, orc_color(s, scattered, depth + 1, u, rays));
        }
        return V(0.0f, 0.0f, 0.0f);
    }
    orc_vec3 ud = unit(r.d);
    float t = 0.5f * (ud.y + 1.0f);
    return add(smul(1.0f - t, V(1.0f, 1.0f, 1.0f)), smul(t, V(0.5f, 0.7f, 1.0f)));
}

void orc_quantise(const float m[3], int out[3])                                          /* main.cpp:136-140 */
{
    for (int k = 0; k < 3; ++k) {
        double v = 255.99 * sqrtf(m[k]);
        out[k] = (v != v) ? INT_MIN : (int)v;     /* x86 cvttsd2si yields INT_MIN for NaN */
    }
}

/* ------------------------------------------------------------------ the reference's lens generator (camera.cpp:33-35)
 * std::mt19937 and std::uniform_real_distribution<float>(0, 1) as libstdc++ 11 implements them (the reference names only
 * the standard classes; MT19937 is Matsumoto & Nishimura's published generator, and the distribution is
 * generate_canonical<float, 24>: one 32-bit draw converted to float, divided by 2^32, and a result that rounds up to 1
 * replaced by the float below 1).  Off by default; orc_lens_mt19937(1, seed) makes the non-keyed sources of orc_render
 * draw their lens samples from it, seeded with the value the reference took from std::random_device. */
static uint32_t g_mt[624]; static int g_mti = 624; static int g_lens_mt = 0;
void orc_lens_mt19937(int enable, uint32_t seed)
{
    g_lens_mt = enable;
    g_mt[0] = seed;
    for (int i = 1; i < 624; ++i) g_mt[i] = 1812433253u * (g_mt[i - 1] ^ (g_mt[i - 1] >> 30)) + (uint32_t)i;
    g_mti = 624;
}
static uint32_t mt_next(void)
{
    if (g_mti >= 624) {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (g_mt[k] & 0x80000000u) | (g_mt[(k + 1) % 624] & 0x7fffffffu);
            g_mt[k] = g_mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g_mti = 0;
    }
    uint32_t y = g_mt[g_mti++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
float orc_lens_uniform(void)
{
    float r = (float)mt_next() / 4294967296.0f;
    return r >= 1.0f ? nextafterf(1.0f, 0.0f) : r;
}

/* ------------------------------------------------------------------ pixel loop (main.cpp:118-141) */
uint64_t orc_render(const orc_scene *s, const orc_camera *cam, int nx, int ny, int ns,
                    orc_usrc *u, int j0, int j1, float *rgb_mean, int *irgb)
{
    uint64_t rays = 0;
    uint64_t lens_state = u->seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    for (int j = j1 - 1; j >= j0; --j) {
        for (int i = 0; i < nx; ++i) {
            orc_vec3 col = V(0.0f, 0.0f, 0.0f);
            for (int smp = 0; smp < ns; ++smp) {
                uint32_t px = (uint32_t)j * (uint32_t)nx + (uint32_t)i;
                if (u->kind == 2) orc_usrc_slot(u, px, (uint32_t)smp, 0);
                float fu = (float)(i + orc_uniform(u)) / (float)nx;
                float fv = (float)(j + orc_uniform(u)) / (float)ny;
                float dx = 0.0f, dy = 0.0f;
                if (cam->lens_radius > 0.0f) {
                    /* RandomUnitInDisk (camera.cpp:31-44).  Keyed source: try 0 is words 2,3 of group 0
                     * of slot (pixel, sample, 0); try t >= 1 is words 0,1 (t odd) or 2,3 (t even) of
                     * group (t + 1) / 2.  Other sources: the reference's own std::mt19937 stream when
                     * orc_lens_mt19937 has given its seed, else a private splitmix stream. */
                    uint32_t t = 0;
                    do {
                        float a, b;
                        if (u->kind == 2) {
                            int hi = (t == 0) || ((t & 1u) == 0);
                            uint32_t grp = (t + 1u) >> 1;
                            a = orc_usrc_word(u, grp, hi ? 2 : 0); b = orc_usrc_word(u, grp, hi ? 3 : 1); ++t;
                        } else if (g_lens_mt) {
                            /* Vec3(dis(gen), dis(gen), 0.0f): the pinned build evaluates the second argument first */
                            b = orc_lens_uniform(); a = orc_lens_uniform();
                        } else {
                            uint64_t z = (lens_state += 0x9E3779B97F4A7C15ull);
                            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                            a = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f); b = (float)((uint32_t)z >> 8) * (1.0f / 16777216.0f);
                        }
                        dx = 2.0f * a - 1.0f; dy = 2.0f * b - 1.0f;
                    } while ((dx * dx) + (dy * dy) + (0.0f * 0.0f) >= 1.0f);
                }
                orc_ray ray = orc_camera_get_ray(cam, fu, fv, dx, dy);
                if (u->kind == 2) orc_usrc_slot(u, px, (uint32_t)smp, 0);
                orc_vec3 c = orc_color(s, ray, 0, u, &rays);
                col = add(col, c);
            }
            float k = 1.0f / (float)ns;                /* Vec3::operator/=(float), vec3.cpp:109-118 */
            col = V(col.x * k, col.y * k, col.z * k);
            size_t o = ((size_t)(j1 - 1 - j) * nx + i) * 3;
            float m[3] = { col.x, col.y, col.z };
            if (rgb_mean) { rgb_mean[o] = m[0]; rgb_mean[o + 1] = m[1]; rgb_mean[o + 2] = m[2]; }
            if (irgb) orc_quantise(m, irgb + o);
        }
    }
    return rays;
}

/* ------------------------------------------------------------------ RandomScene (main.cpp:41-83) */
int orc_random_scene(orc_usrc *u, float *cx, float *cy, float *cz, float *radius, int *type,
                     float *ar, float *ag, float *ab, float *param)
{
    int i = 0;
#define PUT(X, Y, Z, R, T, AR, AG, AB, P) do { cx[i] = (X); cy[i] = (Y); cz[i] = (Z); radius[i] = (R); type[i] = (T); ar[i] = (AR); ag[i] = (AG); ab[i] = (AB); param[i] = (P); ++i; } while (0)
    PUT(0.0f, -1000.0f, 0.0f, 1000.0f, 0, 0.5f, 0.5f, 0.5f, 0.0f);
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            float choose = (float)orc_uniform(u);
            /* right-to-left: the z argument draws before the x argument */
            float z = (float)(b + 0.9f * orc_uniform(u));
            float x = (float)(a + 0.9f * orc_uniform(u));
            orc_vec3 c = V(x, 0.2f, z);
            if (length(sub(c, V(4.0f, 0.2f, 0.0f))) > 0.9f) {
                if (choose < 0.8f) {
                    double b2 = orc_uniform(u), b1 = orc_uniform(u);
                    double g2 = orc_uniform(u), g1 = orc_uniform(u);
                    double r2 = orc_uniform(u), r1 = orc_uniform(u);
                    PUT(x, 0.2f, z, 0.2f, 0, (float)(r1 * r2), (float)(g1 * g2), (float)(b1 * b2), 0.0f);
                } else if (choose < 0.95f) {
                    /* the albedo temporary is built first (its arguments right to left), the fuzz last */
                    float mb = (float)(0.5f * (1.0f + orc_uniform(u)));
                    float mg = (float)(0.5f * (1.0f + orc_uniform(u)));
                    float mr = (float)(0.5f * (1.0f + orc_uniform(u)));
                    float fz = (float)(0.5f * orc_uniform(u));
                    PUT(x, 0.2f, z, 0.2f, 1, mr, mg, mb, orc_metal_fuzz(fz));
                } else {
                    PUT(x, 0.2f, z, 0.2f, 2, 1.0f, 1.0f, 1.0f, 1.5f);
                }
            }
        }
    }
    PUT(0.0f, 1.0f, 0.0f, 1.0f, 2, 1.0f, 1.0f, 1.0f, 1.5f);
    PUT(-4.0f, 1.0f, 0.0f, 1.0f, 0, 0.4f, 0.2f, 0.1f, 0.0f);
    PUT(4.0f, 1.0f, 0.0f, 1.0f, 1, 0.7f, 0.6f, 0.5f, 0.0f);
#undef PUT
    return i;
}
