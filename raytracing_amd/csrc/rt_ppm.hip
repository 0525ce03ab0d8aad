// output.ppm (main.cpp:91-97, 143-147): the ASCII P3 text on the GPU from the device-resident image, and on the host.
#include "rt_internal.hpp"

// ============================================================================ output.ppm on the GPU
// ASCII P3 text of main.cpp:95-97,143-144 ("r g b\n" per pixel) produced from the quantised image
// without a trip through host formatting: a length pass, a scan of per-block lengths, and a write pass
// that lays the block's text out in LDS and streams it to HBM as whole dwords.
constexpr int kPpmBlock = 256, kPpmPxPerThread = 4, kPpmPxPerBlock = kPpmBlock * kPpmPxPerThread;
constexpr int kPpmMaxLine = 36;        // three ints of up to 11 characters, two blanks, one newline

__device__ __forceinline__ int ppm_digits(int32_t v)
{
    const uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    int n = 1;
    if (u >= 10u) n = 2;
    if (u >= 100u) n = 3;
    if (u >= 1000u) {
        n = 4;
        if (u >= 10000u) n = 5;
        if (u >= 100000u) n = 6;
        if (u >= 1000000u) n = 7;
        if (u >= 10000000u) n = 8;
        if (u >= 100000000u) n = 9;
        if (u >= 1000000000u) n = 10;
    }
    return n + (v < 0 ? 1 : 0);
}
__device__ __forceinline__ int ppm_line_length(int32_t r, int32_t g, int32_t b) { return ppm_digits(r) + ppm_digits(g) + ppm_digits(b) + 3; }

// Writes v in decimal followed by `sep` at q (LDS), returns the position after it.
__device__ __forceinline__ char* ppm_put(char* q, int32_t v, char sep)
{
    const int n = ppm_digits(v);
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    if (v < 0) q[0] = '-';
    for (int k = n - 1; k >= (v < 0 ? 1 : 0); --k) { q[k] = (char)('0' + u % 10u); u /= 10u; }
    q[n] = sep;
    return q + n + 1;
}

struct PpmArgs {
    const int32_t* irgb; uint32_t pixels;
    uint32_t* block_len; unsigned long long* block_off;    // [blocks], [blocks + 1]
    uint32_t blocks;
    char* text; int header_len; char header[40];
    // rt_encode_ppm_device_noalloc only: the length goes to device memory and the write pass itself checks the capacity
    long long* d_bytes; long long cap; int check_on_device;
};

__device__ __forceinline__ uint32_t ppm_block_scan(uint32_t mine, uint32_t* s_wave, uint32_t& block_total)
{
    // exclusive scan of `mine` over the 256 threads of the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < kPpmBlock / 64; ++w) { const uint32_t t = s_wave[w]; if (w < wave) base += t; total += t; }
    block_total = total;
    return base + incl - mine;
}

__global__ void __launch_bounds__(kPpmBlock) k_ppm_len(const PpmArgs a)
{
    __shared__ uint32_t s_wave[kPpmBlock / 64];
    const uint32_t first = (blockIdx.x * (uint32_t)kPpmBlock + threadIdx.x) * (uint32_t)kPpmPxPerThread;
    uint32_t mine = 0;
    for (int k = 0; k < kPpmPxPerThread; ++k) {
        const uint32_t px = first + (uint32_t)k;
        if (px < a.pixels) mine += (uint32_t)ppm_line_length(a.irgb[3 * (size_t)px], a.irgb[3 * (size_t)px + 1], a.irgb[3 * (size_t)px + 2]);
    }
    uint32_t total = 0;
    (void)ppm_block_scan(mine, s_wave, total);
    if (threadIdx.x == 0) a.block_len[blockIdx.x] = total;
}

// One block: exclusive scan of the per-block lengths into 64-bit offsets; block_off[blocks] = text bytes after the header.
__global__ void __launch_bounds__(1024) k_ppm_scan(const PpmArgs a)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < a.blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long mine = i < a.blocks ? (unsigned long long)a.block_len[i] : 0ull;
        unsigned long long incl = mine;
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned long long before = s_carry, total = 0ull;
        for (int w = 0; w < 16; ++w) { const unsigned long long t = s_wave[w]; if (w < wave) before += t; total += t; }
        if (i < a.blocks) a.block_off[i] = before + incl - mine;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.block_off[a.blocks] = s_carry; if (a.d_bytes) *a.d_bytes = (long long)s_carry + a.header_len; }
}

__global__ void __launch_bounds__(kPpmBlock) k_ppm_write(const PpmArgs a)
{
    // text of the block, placed so that LDS byte k lands on HBM byte (dst rounded down to 4) + k
    __shared__ __attribute__((aligned(16))) char s_text[kPpmPxPerBlock * kPpmMaxLine + 8];
    __shared__ uint32_t s_wave[kPpmBlock / 64];
    // the same for every thread of the grid: a text that does not fit is not written at all
    if (a.check_on_device && (long long)a.block_off[a.blocks] + a.header_len > a.cap) return;
    const uint32_t first = (blockIdx.x * (uint32_t)kPpmBlock + threadIdx.x) * (uint32_t)kPpmPxPerThread;
    int32_t v[kPpmPxPerThread][3];
    uint32_t mine = 0;
    for (int k = 0; k < kPpmPxPerThread; ++k) {
        const uint32_t px = first + (uint32_t)k;
        if (px < a.pixels) {
            v[k][0] = a.irgb[3 * (size_t)px]; v[k][1] = a.irgb[3 * (size_t)px + 1]; v[k][2] = a.irgb[3 * (size_t)px + 2];
            mine += (uint32_t)ppm_line_length(v[k][0], v[k][1], v[k][2]);
        }
    }
    uint32_t total = 0;
    const uint32_t at = ppm_block_scan(mine, s_wave, total);
    char* dst = a.text + a.header_len + a.block_off[blockIdx.x];
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
    char* q = s_text + shift + at;
    for (int k = 0; k < kPpmPxPerThread; ++k) {
        if (first + (uint32_t)k < a.pixels) { q = ppm_put(q, v[k][0], ' '); q = ppm_put(q, v[k][1], ' '); q = ppm_put(q, v[k][2], '\n'); }
    }
    if (blockIdx.x == 0) { for (int k = threadIdx.x; k < a.header_len; k += kPpmBlock) a.text[k] = a.header[k]; }
    __syncthreads();
    char* dst_al = dst - shift;
    const uint32_t end = shift + total;                  // text occupies LDS bytes [shift, end)
    const uint32_t w_first = (shift + 3u) / 4u, w_last = end / 4u;      // whole dwords [w_first, w_last)
    const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_text);
    uint32_t* g_words = reinterpret_cast<uint32_t*>(dst_al);
    for (uint32_t w = w_first + threadIdx.x; w < w_last; w += kPpmBlock) g_words[w] = s_words[w];
    // bytes before the first whole dword and after the last one (a neighbouring block owns the rest of those dwords)
    if (threadIdx.x < 8) {
        const uint32_t head_end = w_first * 4u < end ? w_first * 4u : end;
        const uint32_t tail_begin = w_last * 4u > head_end ? w_last * 4u : head_end;
        const uint32_t b = threadIdx.x < 4 ? shift + threadIdx.x : tail_begin + (threadIdx.x - 4u);
        const bool ok = threadIdx.x < 4 ? b < head_end : b < end;
        if (ok) dst_al[b] = s_text[b];
    }
}

extern "C" {

// ---------------------------------------------------------------------------- output.ppm
static inline char* put_int(char* q, int32_t v)
{
    char tmp[12]; int n = 0;
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *q++ = '-';
    while (n) *q++ = tmp[--n];
    return q;
}

}   // extern "C"
namespace {
// Decimal text of 0..255 (what the quantiser produces for radiance in [0, 1]): up to three digits in the
// low bytes of a word, so that a value is emitted with one 4-byte store.
struct ByteText {
    uint32_t word[256]; uint8_t len[256];
    ByteText() { for (int v = 0; v < 256; ++v) { char t[8] = { 0 }; len[v] = (uint8_t)(put_int(t, v) - t); memcpy(&word[v], t, 4); } }
};
const ByteText& byte_text() { static const ByteText t; return t; }
inline int int_length(const ByteText& bt, int32_t v) { char t[16]; return (uint32_t)v < 256u ? bt.len[v] : (int)(put_int(t, v) - t); }
}
extern "C" {

int64_t rt_format_ppm(const int32_t* irgb, int width, int height, char* buf, int64_t cap)
{
    if (!irgb || width <= 0 || height <= 0) return 0;
    const ByteText& bt = byte_text();
    const int64_t ints = (int64_t)width * height * 3;
    const int64_t worst = 32 + (int64_t)width * height * 36;   // three ints of up to 11 chars + separators
    char head[48];
    const int hn = snprintf(head, sizeof head, "P3\n%d %d\n255\n", width, height);
    if (!buf || cap < worst) {
        // exact size pass
        int64_t need = hn + ints;
        for (int64_t k = 0; k < ints; ++k) need += int_length(bt, irgb[k]);
        if (!buf || cap < need) return -need;
    }
    char* q = buf; memcpy(q, head, hn); q += hn;
    char* const end = buf + cap;
    auto emit = [&](int32_t v, char sep) {
        if ((uint32_t)v < 256u && end - q >= 8) { memcpy(q, &bt.word[v], 4); q += bt.len[v]; }    // the store may spill up to 3 bytes past the digits
        else q = put_int(q, v);
        *q++ = sep;
    };
    for (int64_t k = 0; k < ints; k += 3) { emit(irgb[k], ' '); emit(irgb[k + 1], ' '); emit(irgb[k + 2], '\n'); }
    return q - buf;
}

int rt_write_ppm(const char* path, const int32_t* irgb, int width, int height)
{
    if (!path || !irgb || width <= 0 || height <= 0) return fail(nullptr, RT_ERR_INVALID, "rt_write_ppm: bad argument");
    const int64_t need = -rt_format_ppm(irgb, width, height, nullptr, 0);
    std::vector<char> buf((size_t)need);
    const int64_t n = rt_format_ppm(irgb, width, height, buf.data(), need);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(nullptr, RT_ERR_INVALID, "rt_write_ppm: cannot open %s", path);
    const size_t w = fwrite(buf.data(), 1, (size_t)n, f);
    fclose(f);
    return w == (size_t)n ? RT_OK : fail(nullptr, RT_ERR_INVALID, "rt_write_ppm: short write to %s", path);
}

int rt_write_ppm_binary(const char* path, const int32_t* irgb, int width, int height)
{
    if (!path || !irgb || width <= 0 || height <= 0) return fail(nullptr, RT_ERR_INVALID, "rt_write_ppm_binary: bad argument");
    char head[48];
    const int hn = snprintf(head, sizeof head, "P6\n%d %d\n255\n", width, height);
    const size_t n = (size_t)width * height * 3;
    std::vector<unsigned char> buf((size_t)hn + n);
    memcpy(buf.data(), head, hn);
    for (size_t k = 0; k < n; ++k) {
        const int32_t v = irgb[k];
        if (v < 0 || v > 255) return fail(nullptr, RT_ERR_INVALID, "rt_write_ppm_binary: value %d at element %zu does not fit a byte (the P3 form writes it as text)", v, k);
        buf[hn + k] = (unsigned char)v;
    }
    FILE* f = fopen(path, "wb");
    if (!f) return fail(nullptr, RT_ERR_INVALID, "rt_write_ppm_binary: cannot open %s", path);
    const size_t w = fwrite(buf.data(), 1, buf.size(), f);
    const int closed = fclose(f);
    return (w == buf.size() && closed == 0) ? RT_OK : fail(nullptr, RT_ERR_INVALID, "rt_write_ppm_binary: short write to %s", path);
}

int rt_encode_ppm_device(rt_context* ctx, const int32_t* d_irgb, int width, int height, char* d_text, int64_t cap, int64_t* bytes, void* stream)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_encode_ppm_device: null context");
    if (!d_irgb || width <= 0 || height <= 0 || !bytes) return fail(ctx, RT_ERR_INVALID, "rt_encode_ppm_device: bad argument");
    const uint64_t pixels64 = (uint64_t)width * (uint64_t)height;
    if (pixels64 > 0x40000000ull) return fail(ctx, RT_ERR_TOO_LARGE, "rt_encode_ppm_device: more than 2^30 pixels");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    PpmArgs a{};
    a.irgb = d_irgb; a.pixels = (uint32_t)pixels64;
    a.blocks = (a.pixels + kPpmPxPerBlock - 1) / kPpmPxPerBlock;
    a.header_len = snprintf(a.header, sizeof a.header, "P3\n%d %d\n255\n", width, height);
    const size_t off_bytes = ((size_t)a.blocks + 1) * sizeof(unsigned long long), need = off_bytes + (size_t)a.blocks * sizeof(uint32_t);
    // The write pass of the call before may still be reading the context's block offsets, possibly on another stream: this call's
    // kernels wait for the event recorded behind it.  (No stream handle is kept: the caller may have destroyed that stream since.)
    if (!ctx->ev_ppm) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_ppm, hipEventDisableTiming));
    if (ctx->ppm_pending) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_ppm, 0));
    if (ctx->ppm_scratch_bytes < need) {
        if (ctx->ppm_pending) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_ppm));       // the old buffer is about to be freed
        (void)hipFree(ctx->d_ppm_scratch); ctx->d_ppm_scratch = nullptr; ctx->ppm_scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_ppm_scratch, need));
        ctx->ppm_scratch_bytes = need;
    }
    a.block_off = static_cast<unsigned long long*>(ctx->d_ppm_scratch);
    a.block_len = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->d_ppm_scratch) + off_bytes);
    a.text = d_text;
    hipLaunchKernelGGL(k_ppm_len, dim3(a.blocks), dim3(kPpmBlock), 0, st, a);
    hipLaunchKernelGGL(k_ppm_scan, dim3(1), dim3(1024), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long body = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&body, a.block_off + a.blocks, sizeof body, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *bytes = (int64_t)body + a.header_len;
    if (!d_text || cap < *bytes) return fail(ctx, RT_ERR_TOO_LARGE, "rt_encode_ppm_device: the text needs %lld bytes, capacity is %lld", (long long)*bytes, (long long)cap);
    hipLaunchKernelGGL(k_ppm_write, dim3(a.blocks), dim3(kPpmBlock), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_ppm, st));
    ctx->ppm_pending = true;
    return RT_OK;
}

// ---------------------------------------------------------------------------- the same encoder without any allocation
// Scratch of rt_encode_ppm_device_noalloc: the per-block text lengths and offsets that rt_encode_ppm_device keeps in
// memory it allocates.  One copy per GPU, at its own address on each, held until the process ends.
}   // extern "C"
constexpr uint32_t kPpmStaticBlocks = RT_PPM_NOALLOC_MAX_PIXELS / kPpmPxPerBlock;
constexpr size_t kPpmStaticOffBytes = ((size_t)kPpmStaticBlocks + 1) * sizeof(unsigned long long);
static __device__ __align__(256) unsigned char scratch[kPpmStaticOffBytes + (size_t)kPpmStaticBlocks * sizeof(uint32_t)];

// Called by rt_create with the context's device current: the device address of this GPU's copy of `scratch`.  In host code
// the array's name is only a handle for this look-up; every later use goes through the pointer found here.
int ppm_static_scratch(rt_context* ctx)
{
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) { ctx->d_ppm_static = nullptr; return fail(ctx, RT_ERR_HIP, "hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e)); }
    ctx->d_ppm_static = ptr;
    return RT_OK;
}
extern "C" {

int rt_encode_ppm_device_noalloc(rt_context* ctx, const int32_t* d_irgb, int width, int height, char* d_text, int64_t cap, int64_t* d_bytes, void* stream)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_encode_ppm_device_noalloc: null context");
    if (!d_irgb || width <= 0 || height <= 0 || !d_bytes || cap < 0) return fail(ctx, RT_ERR_INVALID, "rt_encode_ppm_device_noalloc: bad argument");
    const uint64_t pixels64 = (uint64_t)width * (uint64_t)height;
    if (pixels64 > (uint64_t)RT_PPM_NOALLOC_MAX_PIXELS)
        return fail(ctx, RT_ERR_TOO_LARGE, "rt_encode_ppm_device_noalloc: %llu pixels exceed the %u this entry point is built for; use rt_encode_ppm_device", (unsigned long long)pixels64, (unsigned)RT_PPM_NOALLOC_MAX_PIXELS);
    if (!ctx->d_ppm_static) return fail(ctx, RT_ERR_HIP, "rt_encode_ppm_device_noalloc: the static scratch of this GPU was not found when the context was created");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    PpmArgs a{};
    a.irgb = d_irgb; a.pixels = (uint32_t)pixels64;
    a.blocks = (a.pixels + kPpmPxPerBlock - 1) / kPpmPxPerBlock;
    a.header_len = snprintf(a.header, sizeof a.header, "P3\n%d %d\n255\n", width, height);
    // every word of the scratch that the kernels read is written by this call first (lengths by k_ppm_len, offsets and the
    // total by k_ppm_scan), so what an earlier call left behind does not matter and nothing has to be cleared
    a.block_off = static_cast<unsigned long long*>(ctx->d_ppm_static);
    a.block_len = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->d_ppm_static) + kPpmStaticOffBytes);
    a.text = d_text; a.d_bytes = reinterpret_cast<long long*>(d_bytes); a.cap = (long long)cap; a.check_on_device = 1;
    hipLaunchKernelGGL(k_ppm_len, dim3(a.blocks), dim3(kPpmBlock), 0, st, a);
    hipLaunchKernelGGL(k_ppm_scan, dim3(1), dim3(1024), 0, st, a);
    if (d_text && cap > 0) hipLaunchKernelGGL(k_ppm_write, dim3(a.blocks), dim3(kPpmBlock), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_write_ppm_device(rt_context* ctx, const char* path, const int32_t* d_irgb, int width, int height, void* stream)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_write_ppm_device: null context");
    if (!path) return fail(ctx, RT_ERR_INVALID, "rt_write_ppm_device: null path");
    int64_t bytes = 0;
    // size query first (no text buffer), then the encode into a buffer of exactly that size
    int rc = rt_encode_ppm_device(ctx, d_irgb, width, height, nullptr, 0, &bytes, stream);
    if (rc != RT_ERR_TOO_LARGE || bytes <= 0) return rc == RT_OK ? fail(ctx, RT_ERR_INVALID, "rt_write_ppm_device: empty image") : rc;
    DevBuf text;
    HIP_TRY(ctx, text.up(nullptr, (size_t)bytes + 8));
    rc = rt_encode_ppm_device(ctx, d_irgb, width, height, text.as<char>(), bytes, &bytes, stream);
    if (rc != RT_OK) return rc;
    std::vector<char> host((size_t)bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(ctx, hipMemcpyAsync(host.data(), text.p, (size_t)bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(ctx, RT_ERR_INVALID, "rt_write_ppm_device: cannot open %s", path);
    const size_t w = fwrite(host.data(), 1, host.size(), f);
    const int closed = fclose(f);
    return (w == host.size() && closed == 0) ? RT_OK : fail(ctx, RT_ERR_INVALID, "rt_write_ppm_device: short write to %s", path);
}

}  // extern "C"
