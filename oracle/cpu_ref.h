/* TEST INFRASTRUCTURE — CPU oracle for the render loop.  Plain-C restatement of the reference's
 * arithmetic, operation by operation, so that a build with -ffp-contract=off reproduces the
 * reference's float results bit for bit.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this; the product path never does.
 */
#ifndef ORC_CPU_REF_H
#define ORC_CPU_REF_H
#include <stdint.h>

typedef struct { float x, y, z; } orc_vec3;
typedef struct { orc_vec3 o, d; } orc_ray;
typedef struct { float t; orc_vec3 p, n; int idx; } orc_hit;

/* Scene as structure-of-arrays; type: 0 lambertian, 1 metal, 2 dielectric; param: fuzz | ref_idx. */
typedef struct {
    int n;
    const float *cx, *cy, *cz, *radius;
    const int *type;
    const float *ar, *ag, *ab, *param;
} orc_scene;

typedef struct { orc_vec3 origin, llc, horiz, vert, u, v, w; float lens_radius; } orc_camera;

/* Uniform source.  kind 0: replay of recorded rand() integers; kind 1: glibc random() stream;
 * kind 2: Philox4x32-10 keyed on (pixel, sample, bounce): draw d of a slot is word d%4 of
 * group d/4; value = (word >> 8) / 2^24. */
typedef struct {
    int kind;
    const int *replay; long n_replay; long pos;
    uint64_t seed; uint32_t px, smp, bounce, draw;
    long consumed;
} orc_usrc;

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_uniform(orc_usrc *u);
void orc_usrc_slot(orc_usrc *u, uint32_t px, uint32_t smp, uint32_t bounce);
float orc_usrc_word(const orc_usrc *u, uint32_t group, int w);

int orc_sphere_hit(orc_ray r, orc_vec3 c, float radius, float tmin, float tmax, orc_hit *rec);
int orc_world_hit(const orc_scene *s, orc_ray r, float tmin, float tmax, orc_hit *rec);
orc_vec3 orc_reflect(orc_vec3 v, orc_vec3 n);
int orc_refract(orc_vec3 v, orc_vec3 n, float ni_over_nt, orc_vec3 *out);
float orc_schlick(float cosine, float ref_idx);
orc_vec3 orc_random_in_unit_sphere(orc_usrc *u);
int orc_scatter(int type, orc_vec3 albedo, float param, orc_ray in, const orc_hit *rec, orc_usrc *u, orc_vec3 *att, orc_ray *out);
float orc_metal_fuzz(float fuzz);
void orc_camera_init(orc_camera *c, orc_vec3 from, orc_vec3 at, orc_vec3 up, float vfov, float aspect, float aperture, float focus);
orc_ray orc_camera_get_ray(const orc_camera *c, float s, float t, float disk_x, float disk_y);
orc_vec3 orc_color(const orc_scene *s, orc_ray r, int depth, orc_usrc *u, uint64_t *rays);
/* process-wide counters of orc_color: calls, calls that hit, calls that scattered (main_modern.cpp:23-61) */
void orc_counters_reset(void);
void orc_counters_get(uint64_t out[3]);
void orc_set_max_depth(int d);   /* depth cap of Color(), 50 in the reference */
void orc_quantise(const float mean_rgb[3], int out_rgb[3]);

/* Rows j in [j0, j1) (j counts from the bottom like the reference loop), stored top to bottom.
 * rgb_mean: (j1-j0)*nx*3 linear floats (col / ns); irgb: same shape, int(255.99*sqrt(mean)). */
uint64_t orc_render(const orc_scene *s, const orc_camera *cam, int nx, int ny, int ns,
                    orc_usrc *u, int j0, int j1, float *rgb_mean, int *irgb);

/* Lens samples of the non-keyed sources from the reference's generator, std::mt19937(seed) read through
 * std::uniform_real_distribution<float>(0, 1) (camera.cpp:33-35); orc_lens_uniform is one such draw. */
void orc_lens_mt19937(int enable, uint32_t seed);
float orc_lens_uniform(void);

/* RandomScene() of main.cpp:41-83 drawing from `u`; arrays must hold 501 entries. Returns count. */
int orc_random_scene(orc_usrc *u, float *cx, float *cy, float *cz, float *radius, int *type,
                     float *ar, float *ag, float *ab, float *param);
#endif
