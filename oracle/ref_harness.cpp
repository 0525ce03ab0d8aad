// TEST INFRASTRUCTURE — golden-vector generator and CPU-baseline shim around the UNMODIFIED
// reference sources.  Nothing here is shipped in the product path.
//
// The reference translation unit is included where it lies (-I/root/reference); no reference
// source text is copied into this repository.  Its `main` is renamed so that the reference's
// own Color(), RandomScene(), materials and helpers (main.cpp:13-83, material.h:7-186) become
// callable from this file.  `rand()` is interposed so every uniform the reference consumes can
// be logged, injected, or taken from a counter-based Philox stream keyed on
// (pixel, sample, bounce): that makes the reference's image a pure function of the key, which
// is what the HIP renderer is compared against bit-for-bit.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <random>
#include <unistd.h>

#define main rt_reference_main
#include "main.cpp"
#undef main
#include "appsrc/include/Math/performance.h"   // RenderStats / create_tiles of the multi-threaded variant

extern "C" {
#include "cpu_ref.h"   // only for orc_philox4x32_10 (pinned by Random123 known-answer vectors)
}

// ---------------------------------------------------------------- rand() interposition
enum RandMode { RAND_LIBC, RAND_INJECT, RAND_PHILOX };
static RandMode g_mode = RAND_LIBC;
static std::vector<int> g_inject;
static size_t g_inject_pos = 0;
static bool g_logging = false;
static std::vector<int> g_log;
// Philox key state (RAND_PHILOX): one slot = (pixel, sample, bounce); draw d of a slot is
// word d%4 of group d/4.
static uint32_t g_px = 0, g_smp = 0, g_bounce = 0, g_draw = 0;
static uint64_t g_seed = 0;

extern "C" int rand(void) noexcept
{
    int v;
    if (g_mode == RAND_INJECT) {
        if (g_inject_pos >= g_inject.size()) { fprintf(stderr, "inject underflow\n"); abort(); }
        v = g_inject[g_inject_pos++];
    } else if (g_mode == RAND_PHILOX) {
        uint32_t ctr[4] = { g_px, g_smp, g_bounce, g_draw / 4u }, out[4];
        uint32_t key[2] = { (uint32_t)g_seed, (uint32_t)(g_seed >> 32) };
        orc_philox4x32_10(ctr, key, out);
        v = (int)((out[g_draw % 4u] >> 8) << 7);   // (x>>8)/2^24 == v/2^31 exactly
        ++g_draw;
    } else {
        v = (int)random();
    }
    if (g_logging) g_log.push_back(v);
    return v;
}

// ---------------------------------------------------------------- std::random_device interposition
// Camera::RandomUnitInDisk (camera.cpp:31-44) seeds a thread_local std::mt19937 from std::random_device, so the
// reference's lens samples differ from run to run.  The three out-of-line members of libstdc++'s random_device are
// defined here (the executable's definitions take precedence over the shared library's, as with rand()), so the
// one value the reference draws from it is g_lens_seed and its lens stream becomes a function of that number.
// Nothing of the reference is changed: its own mt19937 / uniform_real_distribution<float> code still runs.
static uint32_t g_lens_seed = 20240607u;
void std::random_device::_M_init(const std::string&) {}
void std::random_device::_M_fini() {}
std::random_device::result_type std::random_device::_M_getval() { return g_lens_seed; }

static void philox_slot(uint32_t px, uint32_t smp, uint32_t bounce) { g_px = px; g_smp = smp; g_bounce = bounce; g_draw = 0; }
static float philox_word(uint32_t px, uint32_t smp, uint32_t bounce, uint32_t group, int w)
{
    uint32_t ctr[4] = { px, smp, bounce, group }, out[4];
    uint32_t key[2] = { (uint32_t)g_seed, (uint32_t)(g_seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
    return (float)(out[w] >> 8) * (1.0f / 16777216.0f);
}

// World wrapper: counts closest-hit queries (= Color() invocations = "rays") and how many of them hit
// (the counter main_modern.cpp:34-37 calls intersection_tests) and, in Philox
// mode, advances the bounce slot: the k-th query of a sample is depth k-1, and the scatter that
// follows it draws from slot bounce = k.
struct CountingWorld : public Hittable {
    Hittable* inner; mutable uint64_t rays = 0, hits = 0;
    explicit CountingWorld(Hittable* w) : inner(w) {}
    bool Hit(const Ray& r, float tmin, float tmax, HitRecord& rec) const override {
        ++rays;
        if (g_mode == RAND_PHILOX) { g_bounce += 1; g_draw = 0; }
        const bool h = inner->Hit(r, tmin, tmax, rec);
        hits += h ? 1u : 0u;
        return h;
    }
};

// ---------------------------------------------------------------- small deterministic generator for test inputs
struct Lcg { uint64_t s; explicit Lcg(uint64_t seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
    float uni() { return (float)(u32() >> 8) * (1.0f / 16777216.0f); }
    float range(float a, float b) { return a + (b - a) * uni(); }
    int draw31() { return (int)(u32() >> 1); } };

static void pv(FILE* f, const Vec3& v) { fprintf(f, " %.9g %.9g %.9g", v.m_fAxis[0], v.m_fAxis[1], v.m_fAxis[2]); }
static Vec3 rand_unit(Lcg& g) { Vec3 p; do { p = Vec3(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)); } while (p.SquaredLength() >= 1.0f || p.SquaredLength() < 1e-4f); return Unit_Vector(p); }

// ---------------------------------------------------------------- scenes
struct FlatSphere { Vec3 c; float r; int type; Vec3 albedo; float param; };
static std::vector<FlatSphere> flatten(Hittable* world)
{
    std::vector<FlatSphere> out;
    HittableList* hl = dynamic_cast<HittableList*>(world);
    for (int i = 0; i < hl->m_iListSize; ++i) {
        Sphere* s = dynamic_cast<Sphere*>(hl->m_oList[i]);
        FlatSphere f; f.c = s->m_oCenter; f.r = s->m_fRadius; f.albedo = Vec3(1, 1, 1); f.param = 0;
        // Material has no type tag and no virtual destructor; typeid on the polymorphic object works.
        if (Lambertian* l = dynamic_cast<Lambertian*>(s->m_oMaterial)) { f.type = 0; f.albedo = l->m_oAlbedo; }
        else if (Metal* m = dynamic_cast<Metal*>(s->m_oMaterial)) { f.type = 1; f.albedo = m->m_oAlbedo; f.param = m->m_fFuzz; }
        else if (Dielectric* d = dynamic_cast<Dielectric*>(s->m_oMaterial)) { f.type = 2; f.param = d->m_fRefIdx; }
        else { fprintf(stderr, "unknown material\n"); abort(); }
        out.push_back(f);
    }
    return out;
}
static void write_scene(const char* path, Hittable* world, long draws)
{
    FILE* f = fopen(path, "w");
    std::vector<FlatSphere> fs = flatten(world);
    fprintf(f, "# cx cy cz radius type(0=lambertian,1=metal,2=dielectric) albedo_r albedo_g albedo_b param(fuzz|ref_idx)\n");
    fprintf(f, "# n=%zu rand_draws_consumed=%ld\n", fs.size(), draws);
    for (const FlatSphere& s : fs) { fprintf(f, "%.9g %.9g %.9g %.9g %d %.9g %.9g %.9g %.9g\n", s.c.m_fAxis[0], s.c.m_fAxis[1], s.c.m_fAxis[2], s.r, s.type, s.albedo.m_fAxis[0], s.albedo.m_fAxis[1], s.albedo.m_fAxis[2], s.param); }
    fclose(f);
}
// Final scene exactly as main() builds it: first rand() consumer of the process, glibc seed 1.
static Hittable* final_scene(long* draws_out)
{
    srand(1);
    g_mode = RAND_LIBC; g_logging = true; g_log.clear();
    Hittable* w = RandomScene();
    g_logging = false; if (draws_out) *draws_out = (long)g_log.size();
    return w;
}
// Four-sphere material scene (performance_comparison.cpp:48-51) plus, optionally, the hollow
// glass shell of main.cpp:107.
static Hittable* three_scene(bool hollow)
{
    Hittable** l = new Hittable*[5]; int n = 0;
    l[n++] = new Sphere(Vec3(0.0f, -100.5f, -1.0f), 100.0f, new Lambertian(Vec3(0.5f, 0.5f, 0.5f)));
    l[n++] = new Sphere(Vec3(0.0f, 0.0f, -1.0f), 0.5f, new Lambertian(Vec3(0.7f, 0.3f, 0.3f)));
    l[n++] = new Sphere(Vec3(-1.0f, 0.0f, -1.0f), 0.5f, new Dielectric(1.5f));
    l[n++] = new Sphere(Vec3(1.0f, 0.0f, -1.0f), 0.5f, new Metal(Vec3(0.8f, 0.6f, 0.2f), 0.3f));
    if (hollow) l[n++] = new Sphere(Vec3(-1.0f, 0.0f, -1.0f), -0.45f, new Dielectric(1.5f));
    return new HittableList(l, n);
}
static Camera scene_camera(const std::string& scene, int nx, int ny, float aperture_override)
{
    if (scene == "final") {
        float ap = aperture_override >= 0 ? aperture_override : 0.1f;
        return Camera(Vec3(13.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f, float(nx) / float(ny), ap, 10.0f);
    }
    float ap = aperture_override >= 0 ? aperture_override : 0.0f;
    return Camera(Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 0.0f, -1.0f), Vec3(0.0f, 1.0f, 0.0f), 90.0f, float(nx) / float(ny), ap, 1.0f);
}
static Hittable* scene_world(const std::string& scene)
{
    if (scene == "final") return final_scene(nullptr);
    if (scene == "three") return three_scene(false);
    if (scene == "three_hollow") return three_scene(true);
    fprintf(stderr, "unknown scene %s\n", scene.c_str()); abort();
}

// ---------------------------------------------------------------- unit vectors
static void gen_sphere_hit(const char* path, int n)
{
    FILE* f = fopen(path, "w"); Lcg g(11);
    fprintf(f, "# ox oy oz dx dy dz cx cy cz r tmin tmax | hit t px py pz nx ny nz\n");
    Lambertian* mat = new Lambertian(Vec3(0.5f, 0.5f, 0.5f));
    for (int k = 0; k < n; ++k) {
        Vec3 o(g.range(-5, 5), g.range(-5, 5), g.range(-5, 5));
        Vec3 c(g.range(-5, 5), g.range(-5, 5), g.range(-5, 5));
        float r = g.range(0.1f, 2.5f);
        int kind = k % 8;
        Vec3 d;
        if (kind == 0) { c = Vec3(0.0f, -1000.0f, 0.0f); r = 1000.0f; o = Vec3(g.range(-12, 13), g.range(0.0f, 3.0f), g.range(-12, 12)); }
        if (kind == 1) r = -r;                                   // hollow shell: normal flips
        if (kind == 2) o = c + rand_unit(g) * (0.7f * fabsf(r));  // origin inside
        // aim roughly at the sphere so that most cases hit; scale direction length
        Vec3 target = c + rand_unit(g) * (fabsf(r) * g.range(0.0f, 1.6f));
        if (kind == 0) target = Vec3(g.range(-12, 12), g.range(-0.5f, 0.5f), g.range(-12, 12));
        d = (target - o) * g.range(0.05f, 3.0f);
        if (kind == 3) d = -d;                                   // sphere behind the origin
        float tmin = 0.001f, tmax = FLT_MAX;
        if (kind == 4) tmax = g.range(0.2f, 3.0f);               // finite closest-so-far
        if (kind == 5) { tmin = g.range(0.5f, 2.0f); }           // first root can fall below tmin
        Ray ray(o, d); Sphere s(c, r, mat); HitRecord rec; rec.m_fT = 0; rec.m_oPoint = Vec3(0, 0, 0); rec.m_oNormal = Vec3(0, 0, 0);
        bool hit = s.Hit(ray, tmin, tmax, rec);
        pv(f, o); pv(f, d); pv(f, c); fprintf(f, " %.9g %.9g %.9g %d %.9g", r, tmin, tmax, hit ? 1 : 0, hit ? rec.m_fT : 0.0f);
        pv(f, hit ? rec.m_oPoint : Vec3(0, 0, 0)); pv(f, hit ? rec.m_oNormal : Vec3(0, 0, 0)); fprintf(f, "\n");
    }
    fclose(f);
}

static void gen_helpers(const char* path, int n)
{
    FILE* f = fopen(path, "w"); Lcg g(23);
    fprintf(f, "# vx vy vz nx ny nz ni_over_nt cosine ref_idx | reflect(3) refract_ok refracted(3) schlick\n");
    for (int k = 0; k < n; ++k) {
        Vec3 v = rand_unit(g) * g.range(0.1f, 4.0f), nrm = rand_unit(g);
        float nint = (k % 3 == 0) ? 1.5f : (k % 3 == 1 ? 1.0f / 1.5f : g.range(0.4f, 2.5f));
        float cosine = g.range(0.0f, 1.0f), ref = (k % 2) ? 1.5f : g.range(1.05f, 2.5f);
        Vec3 refl = Reflect(v, nrm), refr(0, 0, 0);
        bool ok = Refract(v, nrm, nint, refr); if (!ok) refr = Vec3(0, 0, 0);
        float sch = Schlick(cosine, ref);
        pv(f, v); pv(f, nrm); fprintf(f, " %.9g %.9g %.9g", nint, cosine, ref); pv(f, refl); fprintf(f, " %d", ok ? 1 : 0); pv(f, refr); fprintf(f, " %.9g\n", sch);
    }
    fclose(f);
}

static const int NDRAW = 48;
static void gen_scatter(const char* path, int n)
{
    FILE* f = fopen(path, "w"); Lcg g(37);
    fprintf(f, "# type albedo(3) param in_o(3) in_d(3) t p(3) n(3) draws[%d] | ret att(3) out_o(3) out_d(3) ndraws_used\n", NDRAW);
    for (int k = 0; k < n; ++k) {
        int type = k % 3;
        Vec3 albedo(g.uni(), g.uni(), g.uni());
        float param = 0;
        Material* m;
        if (type == 0) m = new Lambertian(albedo);
        else if (type == 1) { param = (k % 9 == 1) ? 0.0f : g.range(0.0f, 1.4f); m = new Metal(albedo, param); }
        else { param = (k % 2) ? 1.5f : g.range(1.1f, 2.4f); m = new Dielectric(param); albedo = Vec3(1, 1, 1); }
        HitRecord rec; rec.m_oPoint = Vec3(g.range(-5, 5), g.range(-5, 5), g.range(-5, 5)); rec.m_oNormal = rand_unit(g); rec.m_fT = g.range(0.01f, 9.0f); rec.m_oMaterial = m;
        if (k % 7 == 3) rec.m_oNormal = -rec.m_oNormal;
        Vec3 d = rand_unit(g) * g.range(0.05f, 3.0f);
        if (type == 2 && k % 4 == 0) {   // grazing incidence from inside: forces total internal reflection cases
            Vec3 tang = Unit_Vector(Cross(rec.m_oNormal, rand_unit(g)));
            d = (tang + rec.m_oNormal * g.range(0.01f, 0.5f)) * g.range(0.1f, 2.0f);
        }
        Vec3 o = rec.m_oPoint - d * rec.m_fT;
        Ray in(o, d);
        g_inject.clear(); for (int i = 0; i < NDRAW; ++i) g_inject.push_back(g.draw31());
        g_inject_pos = 0; g_mode = RAND_INJECT;
        Vec3 att(0, 0, 0); Ray out(Vec3(0, 0, 0), Vec3(0, 0, 0));
        bool ret = m->Scatter(in, rec, att, out);
        g_mode = RAND_LIBC;
        float stored_param = param; if (type == 1) stored_param = static_cast<Metal*>(m)->m_fFuzz;   // ctor clamp (material.h:73-83)
        fprintf(f, "%d", type); pv(f, albedo); fprintf(f, " %.9g", param); pv(f, o); pv(f, d); fprintf(f, " %.9g", rec.m_fT); pv(f, rec.m_oPoint); pv(f, rec.m_oNormal);
        for (int i = 0; i < NDRAW; ++i) fprintf(f, " %d", g_inject[i]);
        fprintf(f, " %d", ret ? 1 : 0); pv(f, att); pv(f, out.m_oOrigin); pv(f, out.m_oDirection); fprintf(f, " %zu %.9g\n", g_inject_pos, stored_param);
    }
    fclose(f);
}

static void gen_camera(const char* path)
{
    FILE* f = fopen(path, "w"); Lcg g(41);
    fprintf(f, "# kind=ctor: from(3) at(3) up(3) vfov aspect aperture focus | origin llc horiz vert u v w lens_radius\n");
    fprintf(f, "# kind=ray:  (same ctor args) s t | ray_o(3) ray_d(3)   (aperture>0 rows use the reference's own lens RNG)\n");
    struct Cfg { Vec3 from, at, up; float vfov, aspect, ap, focus; };
    std::vector<Cfg> cfgs = {
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 1200.0f / 800.0f, 0.1f, 10.0f },
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 400.0f / 300.0f, 0.1f, 10.0f },
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 3840.0f / 2160.0f, 0.1f, 10.0f },
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 1920.0f / 1080.0f, 0.0f, 10.0f },
        { Vec3(0, 0, 0), Vec3(0, 0, -1), Vec3(0, 1, 0), 90.0f, 400.0f / 225.0f, 0.0f, 1.0f },
        { Vec3(-2, 2, 1), Vec3(0, 0, -1), Vec3(0, 1, 0), 35.0f, 2.0f, 2.0f, 3.4f },
        { Vec3(3, 3, 2), Vec3(0, 0, -1), Vec3(0.2f, 1, 0.1f), 55.0f, 1.0f, 0.5f, 5.2f },
    };
    for (const Cfg& c : cfgs) {
        Camera cam(c.from, c.at, c.up, c.vfov, c.aspect, c.ap, c.focus);
        fprintf(f, "ctor"); pv(f, c.from); pv(f, c.at); pv(f, c.up); fprintf(f, " %.9g %.9g %.9g %.9g", c.vfov, c.aspect, c.ap, c.focus);
        pv(f, cam.m_oOrigin); pv(f, cam.m_oLowerLeftCorner); pv(f, cam.m_oHorizontal); pv(f, cam.m_oVertical); pv(f, cam.m_oU); pv(f, cam.m_oV); pv(f, cam.m_oW); fprintf(f, " %.9g\n", cam.m_fLensRadius);
        for (int k = 0; k < 24; ++k) {
            float s = g.uni(), t = g.uni();
            Ray r = cam.GetRay(s, t);
            fprintf(f, "ray"); pv(f, c.from); pv(f, c.at); pv(f, c.up); fprintf(f, " %.9g %.9g %.9g %.9g %.9g %.9g", c.vfov, c.aspect, c.ap, c.focus, s, t);
            pv(f, r.m_oOrigin); pv(f, r.m_oDirection); fprintf(f, "\n");
        }
    }
    fclose(f);
}

// Camera::GetRay with the lens in use (camera.cpp:31-53), called on the reference's own Camera objects.  The disk
// sample each call used is recovered exactly: a second std::mt19937 seeded with the same value runs beside the
// reference's thread_local one through the same rejection loop, and of the two ways its two draws can be assigned
// to (x, y) -- the language leaves the order of the two dis(gen) arguments open -- the one that reproduces the
// returned origin bit for bit is written out.  Must be the first user of GetRay in its process.
static void gen_camera_dof(const char* path)
{
    FILE* f = fopen(path, "w"); Lcg g(67);
    std::mt19937 mirror(g_lens_seed);
    std::uniform_real_distribution<float> dis(0.0f, 1.0f);
    struct Cfg { Vec3 from, at, up; float vfov, aspect, ap, focus; };
    std::vector<Cfg> cfgs = {
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 1200.0f / 800.0f, 0.1f, 10.0f },
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 3840.0f / 2160.0f, 0.1f, 10.0f },
        { Vec3(13, 2, 3), Vec3(0, 0, 0), Vec3(0, 1, 0), 20.0f, 400.0f / 300.0f, 0.1f, 10.0f },
        { Vec3(-2, 2, 1), Vec3(0, 0, -1), Vec3(0, 1, 0), 35.0f, 2.0f, 2.0f, 3.4f },
        { Vec3(3, 3, 2), Vec3(0, 0, -1), Vec3(0.2f, 1, 0.1f), 55.0f, 1.0f, 0.5f, 5.2f },
    };
    std::vector<std::string> rows; int first_is_x = 0, first_is_y = 0;
    for (const Cfg& c : cfgs) {
        Camera cam(c.from, c.at, c.up, c.vfov, c.aspect, c.ap, c.focus);
        for (int k = 0; k < 40; ++k) {
            float s = g.uni(), t = g.uni();
            Ray r = cam.GetRay(s, t);
            float a, b; int tries = 0;
            do { a = dis(mirror); b = dis(mirror); ++tries; } while ((2.0f * a - 1.0f) * (2.0f * a - 1.0f) + (2.0f * b - 1.0f) * (2.0f * b - 1.0f) + 0.0f >= 1.0f);
            float disk[2] = { 0, 0 }; bool found = false;
            for (int order = 0; order < 2 && !found; ++order) {
                const float x = 2.0f * (order == 0 ? a : b) - 1.0f, y = 2.0f * (order == 0 ? b : a) - 1.0f;
                Vec3 rd = cam.m_fLensRadius * Vec3(x, y, 0.0f);
                Vec3 o = cam.m_oOrigin + (cam.m_oU * rd.GetX() + cam.m_oV * rd.GetY());
                if (memcmp(o.m_fAxis, r.m_oOrigin.m_fAxis, sizeof o.m_fAxis) == 0) { disk[0] = x; disk[1] = y; found = true; (order == 0 ? first_is_x : first_is_y) += (a != b); }
            }
            if (!found) { fprintf(stderr, "dof: lens sample not recovered (config %zu, ray %d)\n", (size_t)(&c - &cfgs[0]), k); abort(); }
            char buf[1024]; int n = 0;
            auto put3 = [&](const Vec3& v) { n += snprintf(buf + n, sizeof buf - n, " %.9g %.9g %.9g", v.m_fAxis[0], v.m_fAxis[1], v.m_fAxis[2]); };
            n += snprintf(buf + n, sizeof buf - n, "ray"); put3(c.from); put3(c.at); put3(c.up);
            n += snprintf(buf + n, sizeof buf - n, " %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d", c.vfov, c.aspect, c.ap, c.focus, s, t, disk[0], disk[1], tries);
            put3(r.m_oOrigin); put3(r.m_oDirection);
            rows.push_back(buf);
        }
    }
    if (first_is_x != 0 && first_is_y != 0) { fprintf(stderr, "dof: inconsistent draw order\n"); abort(); }
    fprintf(f, "# Camera::GetRay of the reference with std::random_device pinned to %u (camera.cpp:31-53)\n", g_lens_seed);
    fprintf(f, "# first draw of a lens try is %s\n", first_is_y ? "y" : "x");
    fprintf(f, "# ray: from(3) at(3) up(3) vfov aspect aperture focus s t disk_x disk_y tries | ray_o(3) ray_d(3)\n");
    for (const std::string& r : rows) fprintf(f, "%s\n", r.c_str());
    fclose(f);
}

// Whole-path vectors: Color() on the final scene for camera rays (aperture 0 => no lens RNG),
// with every rand() draw logged so a replay reproduces the path exactly.
static void gen_paths(const char* path, int n)
{
    FILE* f = fopen(path, "w"); Lcg g(53);
    CountingWorld world(final_scene(nullptr));
    Camera cam = scene_camera("final", 400, 300, 0.0f);
    fprintf(f, "# ray_o(3) ray_d(3) | r g b rays ndraws draws...\n");
    for (int k = 0; k < n; ++k) {
        float s = g.uni(), t = (k % 3 == 0) ? g.range(0.0f, 0.35f) : g.uni();   // bias a third toward the sphere field
        Ray r = cam.GetRay(s, t);
        srand(1000 + k); g_mode = RAND_LIBC; g_logging = true; g_log.clear(); world.rays = 0;
        Vec3 c = Color(r, &world, 0);
        g_logging = false;
        pv(f, r.m_oOrigin); pv(f, r.m_oDirection); pv(f, c); fprintf(f, " %llu %zu", (unsigned long long)world.rays, g_log.size());
        for (int v : g_log) fprintf(f, " %d", v);
        fprintf(f, "\n");
    }
    fclose(f);
}

// ---------------------------------------------------------------- renders
// mode "libc":   the loop of main.cpp:118-141 as written there (rand() jitter, Camera::GetRay with
//                its own lens RNG), glibc stream seeded `seed` after the scene is built.
// mode "philox": same loop, every uniform taken from the keyed Philox stream.  The lens sample is
//                drawn from slot (pixel, sample, 0) (words named in the loop below) and applied with the expression of
//                camera.cpp:48-52 on the reference Camera's public fields, because the reference's
//                own lens generator (camera.cpp:33-35) cannot be seeded.
struct RenderOut { std::vector<float> rgb; uint64_t rays, hits; };
static RenderOut render(const std::string& scene, int nx, int ny, int ns, const std::string& mode, uint64_t seed, int j0, int j1, float aperture_override)
{
    srand(1);   // fresh-process generator state
    Hittable* base = scene_world(scene);
    CountingWorld world(base);
    Camera cam = scene_camera(scene, nx, ny, aperture_override);
    // libc mode: seed 0 continues the stream the scene construction left behind, exactly as
    // main() does; any other seed reseeds after the scene is built.
    if (mode == "libc") { g_mode = RAND_LIBC; if (seed != 0) srand((unsigned)seed); }
    else { g_mode = RAND_PHILOX; g_seed = seed; }
    RenderOut out; out.rgb.assign((size_t)(j1 - j0) * nx * 3, 0.0f);
    for (int j = j1 - 1; j >= j0; --j) {
        for (int i = 0; i < nx; ++i) {
            Vec3 col(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < ns; ++s) {
                Ray ray;
                if (mode == "libc") {
                    float u = float(i + (std::rand() / (RAND_MAX + 1.0))) / float(nx);
                    float v = float(j + (std::rand() / (RAND_MAX + 1.0))) / float(ny);
                    ray = cam.GetRay(u, v);
                } else {
                    uint32_t px = (uint32_t)j * (uint32_t)nx + (uint32_t)i;
                    philox_slot(px, (uint32_t)s, 0);
                    float u = float(i + (std::rand() / (RAND_MAX + 1.0))) / float(nx);
                    float v = float(j + (std::rand() / (RAND_MAX + 1.0))) / float(ny);
                    if (cam.m_fLensRadius > 0.0f) {
                        // lens try t: words 2,3 of group 0 for t = 0, then words 0,1 (t odd) / 2,3 (t even) of group (t+1)/2
                        Vec3 p; uint32_t t = 0;
                        do {
                            const bool hi = (t == 0) || ((t & 1u) == 0); const uint32_t grp = (t + 1u) >> 1;
                            p = 2.0f * Vec3(philox_word(px, s, 0, grp, hi ? 2 : 0), philox_word(px, s, 0, grp, hi ? 3 : 1), 0.0f) - Vec3(1.0f, 1.0f, 0.0f); ++t;
                        } while (Dot(p, p) >= 1.0f);
                        Vec3 rd = cam.m_fLensRadius * p;
                        Vec3 offset = cam.m_oU * rd.GetX() + cam.m_oV * rd.GetY();
                        ray = Ray(cam.m_oOrigin + offset, cam.m_oLowerLeftCorner + u * cam.m_oHorizontal + v * cam.m_oVertical - cam.m_oOrigin - offset);
                    } else {
                        ray = cam.GetRay(u, v);   // lens radius 0: the lens sample is multiplied by zero
                    }
                    philox_slot(px, (uint32_t)s, 0);   // CountingWorld bumps the slot to 1 on the first query
                }
                col += Color(ray, &world, 0);
            }
            col /= float(ns);
            float* o = &out.rgb[((size_t)(j1 - 1 - j) * nx + i) * 3];   // rows stored top -> bottom like the PPM
            o[0] = col[0]; o[1] = col[1]; o[2] = col[2];
        }
    }
    g_mode = RAND_LIBC;
    out.rays = world.rays; out.hits = world.hits;
    return out;
}

// Timing entry for the CPU baseline (libc mode, rows [j0,j1)): returns rays traced.
extern "C" uint64_t ref_render_rows(const char* scene, int nx, int ny, int ns, unsigned seed, int j0, int j1, float* rgb_out)
{
    RenderOut r = render(scene, nx, ny, ns, "libc", seed, j0, j1, -1.0f);
    if (rgb_out) memcpy(rgb_out, r.rgb.data(), r.rgb.size() * sizeof(float));
    return r.rays;
}

#ifndef REF_HARNESS_NO_MAIN
int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness scene|units|dof|mainppm|tiles|stats|paths|render ...\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "scene") { long draws = 0; Hittable* w = final_scene(&draws); write_scene(argv[2], w, draws); return 0; }
    if (cmd == "units") {
        std::string d = argv[2];
        gen_sphere_hit((d + "/sphere_hit.txt").c_str(), 1536);
        gen_helpers((d + "/helpers.txt").c_str(), 600);
        gen_scatter((d + "/scatter.txt").c_str(), 600);
        gen_camera((d + "/camera.txt").c_str());
        return 0;
    }
    if (cmd == "tiles") {
        // create_tiles (performance.h:143-155) for a few image sizes: "w h samples tile_size count" then one "x y w h s" line per tile
        FILE* f = fopen(argv[2], "w");
        const int cases[][4] = { { 2560, 1440, 10, 64 }, { 1200, 800, 10, 64 }, { 100, 70, 3, 32 }, { 64, 64, 1, 64 }, { 65, 1, 2, 64 }, { 7, 200, 5, 16 } };
        for (const auto& c : cases) {
            const std::vector<RenderTile> t = create_tiles(c[0], c[1], c[2], c[3]);
            fprintf(f, "%d %d %d %d %zu\n", c[0], c[1], c[2], c[3], t.size());
            for (const RenderTile& r : t) fprintf(f, "%d %d %d %d %d\n", r.x_start, r.y_start, r.width, r.height, r.samples_per_pixel);
        }
        fclose(f);
        return 0;
    }
    if (cmd == "stats") {
        // RenderStats::print_stats (performance.h:51-59) on fixed numbers, to stdout
        RenderStats st; st.rays_traced = 98765432; st.intersection_tests = 61234567; st.material_evaluations = 51234567; st.total_time_ms = 1234.5678;
        st.print_stats();
        return 0;
    }
    if (cmd == "dof") { gen_camera_dof(argv[2]); return 0; }
    if (cmd == "mainppm") {
        // The reference's own main() (main.cpp:85-149), unmodified, run once in directory argv[2]: 400x300, 5 samples, glibc's
        // default rand() stream, lens generator seeded with g_lens_seed.  output.ppm is created empty beforehand because
        // main.cpp:91-93 opens it with std::fstream's default mode, which does not create a missing file.
        if (chdir(argv[2]) != 0) { perror("chdir"); return 2; }
        { FILE* f = fopen("output.ppm", "w"); if (!f) { perror("output.ppm"); return 2; } fclose(f); }
        g_mode = RAND_LIBC;
        return rt_reference_main(0, nullptr);
    }
    if (cmd == "paths") { gen_paths(argv[2], atoi(argv[3])); return 0; }
    if (cmd == "render") {
        // render <scene> <nx> <ny> <ns> <libc|philox> <seed> <j0> <j1> <aperture|-1> <out.f32>
        if (argc < 12) { fprintf(stderr, "render: missing args\n"); return 2; }
        RenderOut r = render(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argv[6], strtoull(argv[7], 0, 10), atoi(argv[8]), atoi(argv[9]), (float)atof(argv[10]));
        FILE* f = fopen(argv[11], "wb"); fwrite(r.rgb.data(), sizeof(float), r.rgb.size(), f); fclose(f);
        printf("rays %llu hits %llu\n", (unsigned long long)r.rays, (unsigned long long)r.hits);
        return 0;
    }
    fprintf(stderr, "unknown command\n"); return 2;
}
#endif
