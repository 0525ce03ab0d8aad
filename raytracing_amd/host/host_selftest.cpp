// Checks the host mirror against the vectors the reference's own classes produced (tests/golden/):
// RandomScene() + Flatten vs scene_final.txt, Camera vs camera.txt, Sphere::Hit vs sphere_hit.txt,
// the three Scatter functions vs scatter.txt (rand() interposed to replay the recorded draws).
// Build with -ffp-contract=off; every comparison is bit-for-bit.  Prints "ok <n>" per suite.
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "gpu_render.h"
#include "performance.h"
#include "scene.h"

static std::vector<int> g_replay; static size_t g_pos = 0; static bool g_use_replay = false;
extern "C" int rand(void) noexcept
{
    if (g_use_replay) { if (g_pos >= g_replay.size()) { std::fprintf(stderr, "replay underflow\n"); std::abort(); } return g_replay[g_pos++]; }
    return (int)random();
}

static bool same(float a, float b) { return a == b || (a != a && b != b); }
static std::vector<std::vector<std::string>> rows(const std::string& path)
{
    std::vector<std::vector<std::string>> out; std::ifstream f(path); std::string line;
    while (std::getline(f, line)) { if (line.empty() || line[0] == '#') continue; std::istringstream ss(line); std::vector<std::string> r; std::string w; while (ss >> w) r.push_back(w); out.push_back(r); }
    return out;
}
static float F(const std::string& s) { return std::strtof(s.c_str(), nullptr); }
static Vec3 V(const std::vector<std::string>& r, int o) { return Vec3(F(r[o]), F(r[o + 1]), F(r[o + 2])); }
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv)
{
    const std::string g = argc > 1 ? argv[1] : "tests/golden";
    {   // scene
        srand(1);
        Hittable* w = RandomScene();
        const rtamd::FlatScene f = rtamd::Flatten(w);
        const auto t = rows(g + "/scene_final.txt");
        REQUIRE((int)t.size() == f.size() && f.size() == 487);
        for (int i = 0; i < f.size(); ++i) {
            REQUIRE(same(f.cx[i], F(t[i][0])) && same(f.cy[i], F(t[i][1])) && same(f.cz[i], F(t[i][2])) && same(f.radius[i], F(t[i][3])));
            REQUIRE(f.material[i] == std::atoi(t[i][4].c_str()));
            REQUIRE(same(f.albedo[3 * i], F(t[i][5])) && same(f.albedo[3 * i + 1], F(t[i][6])) && same(f.albedo[3 * i + 2], F(t[i][7])) && same(f.param[i], F(t[i][8])));
        }
        DestroyScene(w);
        std::printf("ok scene %d\n", f.size());
    }
    {   // camera
        int n = 0;
        for (const auto& r : rows(g + "/camera.txt")) {
            const Camera c(V(r, 1), V(r, 4), V(r, 7), F(r[10]), F(r[11]), F(r[12]), F(r[13]));
            if (r[0] == "ctor") {
                const Vec3 want[7] = { V(r, 14), V(r, 17), V(r, 20), V(r, 23), V(r, 26), V(r, 29), V(r, 32) };
                const Vec3 got[7] = { c.m_oOrigin, c.m_oLowerLeftCorner, c.m_oHorizontal, c.m_oVertical, c.m_oU, c.m_oV, c.m_oW };
                for (int k = 0; k < 7; ++k) for (int a = 0; a < 3; ++a) REQUIRE(same(got[k][a], want[k][a]));
                REQUIRE(same(c.m_fLensRadius, F(r[35])));
                const rt_camera_pod pod = rtamd::ToPod(c);
                REQUIRE(same(pod.lower_left_corner[2], want[1][2]) && same(pod.lens_radius, c.m_fLensRadius));
                ++n;
            } else if (c.m_fLensRadius == 0.0f) {
                const Ray ray = c.GetRay(F(r[14]), F(r[15]));
                for (int a = 0; a < 3; ++a) REQUIRE(same(ray.m_oOrigin[a], F(r[16 + a])) && same(ray.m_oDirection[a], F(r[19 + a])));
                ++n;
            }
        }
        std::printf("ok camera %d\n", n);
    }
    {   // Sphere::Hit
        int n = 0; Lambertian mat(Vec3(0.5f, 0.5f, 0.5f));
        for (const auto& r : rows(g + "/sphere_hit.txt")) {
            const Sphere s(V(r, 6), F(r[9]), &mat); HitRecord rec;
            const bool hit = s.Hit(Ray(V(r, 0), V(r, 3)), F(r[10]), F(r[11]), rec);
            REQUIRE(hit == (std::atoi(r[12].c_str()) != 0));
            if (hit) { REQUIRE(same(rec.m_fT, F(r[13]))); for (int a = 0; a < 3; ++a) REQUIRE(same(rec.m_oPoint[a], F(r[14 + a])) && same(rec.m_oNormal[a], F(r[17 + a]))); REQUIRE(rec.m_oMaterial == &mat); }
            ++n;
        }
        std::printf("ok sphere_hit %d\n", n);
    }
    {   // Scatter
        int n = 0;
        for (const auto& r : rows(g + "/scatter.txt")) {
            const int type = std::atoi(r[0].c_str());
            Material* m = type == 0 ? (Material*)new Lambertian(V(r, 1)) : type == 1 ? (Material*)new Metal(V(r, 1), F(r[4])) : (Material*)new Dielectric(F(r[4]));
            HitRecord rec; rec.m_fT = F(r[11]); rec.m_oPoint = V(r, 12); rec.m_oNormal = V(r, 15); rec.m_oMaterial = m;
            g_replay.clear(); for (int k = 0; k < 48; ++k) g_replay.push_back(std::atoi(r[18 + k].c_str()));
            g_pos = 0; g_use_replay = true;
            Vec3 att(0, 0, 0); Ray out(Vec3(0, 0, 0), Vec3(0, 0, 0));
            const bool ret = m->Scatter(Ray(V(r, 5), V(r, 8)), rec, att, out);
            g_use_replay = false;
            REQUIRE(ret == (std::atoi(r[66].c_str()) != 0));
            REQUIRE((int)g_pos == std::atoi(r[76].c_str()));
            for (int a = 0; a < 3; ++a) REQUIRE(same(att[a], F(r[67 + a])) && same(out.m_oOrigin[a], F(r[70 + a])) && same(out.m_oDirection[a], F(r[73 + a])));
            delete m; ++n;
        }
        std::printf("ok scatter %d\n", n);
    }
    {   // create_tiles (performance.h:143-155)
        const auto t = rows(g + "/tiles.txt");
        size_t at = 0; int cases = 0;
        while (at < t.size()) {
            const auto& h = t[at++];
            REQUIRE(h.size() == 5);
            const std::vector<RenderTile> tiles = create_tiles(std::atoi(h[0].c_str()), std::atoi(h[1].c_str()), std::atoi(h[2].c_str()), std::atoi(h[3].c_str()));
            REQUIRE(tiles.size() == (size_t)std::atol(h[4].c_str()) && at + tiles.size() <= t.size());
            for (const RenderTile& r : tiles) {
                const auto& w = t[at++];
                REQUIRE(r.x_start == std::atoi(w[0].c_str()) && r.y_start == std::atoi(w[1].c_str()) && r.width == std::atoi(w[2].c_str()) &&
                        r.height == std::atoi(w[3].c_str()) && r.samples_per_pixel == std::atoi(w[4].c_str()));
            }
            ++cases;
        }
        std::printf("ok tiles %d\n", cases);
    }
    {   // RenderStats::print_stats (performance.h:51-59): the caller compares what follows with the reference's text
        RenderStats st; st.rays_traced = 98765432; st.intersection_tests = 61234567; st.material_evaluations = 51234567; st.total_time_ms = 1234.5678;
        std::printf("stats report follows\n");
        st.print_stats();
        st.reset();
        REQUIRE(st.rays_traced.load() == 0 && st.total_time_ms == 0.0);
        PerformanceTimer timer; timer.begin();
        REQUIRE(timer.elapsed_ms() >= 0.0 && timer.elapsed_seconds() < 60.0);
    }
    return 0;
}
