// C ABI of libmdisp.so: contexts, host-buffer entry points (upload -> kernels -> download) and
// device-resident entry points.  See include/mdisp.h for the contract of every function.
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <new>

static thread_local char g_err[512] = "";

void mdisp_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* mdisp_last_error(void) { return g_err; }

// the scheduling overrides are read by compute paths on any host thread and written by mdisp_tuning_set: atomic words
static std::atomic<int> g_tune[MDISP_TUNE_COUNT];
static const bool g_tune_ready = [] { for (auto& v : g_tune) v.store(-1, std::memory_order_relaxed); return true; }();      // -1: the built-in choice
int md_tune(int key) { return key >= 0 && key < MDISP_TUNE_COUNT ? g_tune[key].load(std::memory_order_relaxed) : -1; }
void md_tune_clear(int key) { if (key >= 0 && key < MDISP_TUNE_COUNT) g_tune[key].store(-1, std::memory_order_relaxed); }
// reads a one-shot override and clears it in one step, so two threads cannot both consume it
int md_tune_take(int key) { return key >= 0 && key < MDISP_TUNE_COUNT ? g_tune[key].exchange(-1, std::memory_order_relaxed) : -1; }
extern "C" int mdisp_tuning_set(int key, int value)
{
    MD_REQUIRE(key >= 0 && key < MDISP_TUNE_COUNT, MDISP_E_BADARG, "unknown tuning key %d", key);
    g_tune[key].store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return MDISP_OK;
}
extern "C" int mdisp_tuning_get(int key) { return md_tune(key); }
extern "C" int mdisp_version(void) { return 100; }

extern "C" int mdisp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Scratch of mdisp_census_stereo_dev_noalloc, the entry point that allocates nothing: a file-scope device array that
// every GPU the library runs on holds until the process ends.  In host code the array's name is only a stand-in for
// the symbol; its address on a device comes from hipGetSymbolAddress with that device current (every GPU has its own
// copy at its own address) and only that pointer is ever handed to kernels.
static __device__ __align__(256) unsigned char scratch[MDISP_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static u8* g_scratch_ptr[64] = {};
// `device` must be the current device
static int scratch_of_device(int device, u8** out)
{
    MD_REQUIRE(device >= 0 && device < 64, MDISP_E_BADARG, "device %d beyond the 64 the scratch table holds", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        MD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        MD_REQUIRE(p, MDISP_E_HIP, "hipGetSymbolAddress returned a null scratch address");
        g_scratch_ptr[device] = (u8*)p;
    }
    *out = g_scratch_ptr[device];
    return MDISP_OK;
}

static int create_ctx(mdisp_ctx** out, int device, void* stream, bool useGiven)
{
    MD_REQUIRE(out, MDISP_E_BADARG, "null context pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mdisp_set_error("no HIP device available; libmdisp has no CPU fallback");
        return MDISP_E_NODEVICE;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    MD_REQUIRE(device < n, MDISP_E_BADARG, "device %d out of range (%d present)", device, n);
    if (hipSetDevice(device) != hipSuccess) { mdisp_set_error("hipSetDevice(%d) failed", device); return MDISP_E_NODEVICE; }
    mdisp_ctx* c = new (std::nothrow) mdisp_ctx();
    MD_REQUIRE(c, MDISP_E_NOMEM, "out of host memory");
    c->device = device; c->ws = nullptr; c->ws_bytes = 0; c->timing = false; c->sgm_err = nullptr; c->sgm_pending = false; c->aux = nullptr; c->aux_ready = false; c->lanes_ready = false; c->stage = nullptr; c->stage_bytes = 0;
    for (int i = 0; i < 3; i++) c->ev_valid[i] = false;
    if (useGiven) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c; mdisp_set_error("hipStreamCreate failed"); return MDISP_E_HIP;
        }
        c->own_stream = true;
    }
    for (int i = 0; i < 6; i++)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { mdisp_set_error("hipEventCreate failed"); delete c; return MDISP_E_HIP; }
    // the sticky hand-off error word of the pipelined SGM kernels (one 256-byte line of its own)
    if (hipMalloc((void**)&c->sgm_err, 256) != hipSuccess || hipMemset(c->sgm_err, 0, 256) != hipSuccess) {
        mdisp_set_error("hipMalloc of the context's status word failed"); delete c; return MDISP_E_NOMEM;
    }
    // the no-allocation entry point's scratch address: resolved here when possible (hipGetSymbolAddress allocates nothing);
    // a failure only disables that one entry point, which then tries again and reports the error itself
    if (scratch_of_device(device, &c->noalloc_scratch) != MDISP_OK) c->noalloc_scratch = nullptr;
    *out = c;
    return MDISP_OK;
}

extern "C" int mdisp_create(mdisp_ctx** out, int device, void* stream) { return create_ctx(out, device, stream, stream != nullptr); }
extern "C" int mdisp_create_on_stream(mdisp_ctx** out, int device, void* stream) { return create_ctx(out, device, stream, true); }

extern "C" void mdisp_destroy(mdisp_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ws) (void)hipFree(c->ws);
    for (int i = 0; i < 6; i++) (void)hipEventDestroy(c->ev[i]);
    if (c->aux_ready) {
        (void)hipStreamSynchronize(c->aux);
        for (int i = 0; i < 3; i++) (void)hipEventDestroy(c->aux_ev[i]);
        (void)hipStreamDestroy(c->aux);
    }
    if (c->lanes_ready) {
        for (int i = 0; i < 3; i++) { (void)hipStreamSynchronize(c->lane[i]); (void)hipStreamDestroy(c->lane[i]); }
        for (int i = 0; i < 4; i++) (void)hipEventDestroy(c->lane_ev[i]);
    }
    if (c->stage) (void)hipFree(c->stage);
    if (c->sgm_err) (void)hipFree(c->sgm_err);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mdisp_synchronize(mdisp_ctx* c)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    MD_HIP(hipStreamSynchronize(c->stream));
    return md_check_sgm_error(c);
}

int md_ctx_aux(mdisp_ctx* c, md_aux_streams* out)
{
    if (!c->aux_ready) {
        MD_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
        for (int i = 0; i < 3; i++) MD_HIP(hipEventCreateWithFlags(&c->aux_ev[i], hipEventDisableTiming));
        c->aux_ready = true;
    }
    out->aux = c->aux;
    for (int i = 0; i < 3; i++) out->ev[i] = c->aux_ev[i];
    return MDISP_OK;
}

int md_check_sgm_error(mdisp_ctx* c)
{
    if (!c->sgm_pending) return MDISP_OK;
    c->sgm_pending = false;
    int f = 0;
    MD_HIP(hipMemcpy(&f, c->sgm_err, sizeof(int), hipMemcpyDeviceToHost));
    if (f) {
        // the word is sticky on the device (no call clears it, so no later call can hide it): clear it only here
        MD_HIP(hipMemset(c->sgm_err, 0, sizeof(int)));
        mdisp_set_error("SGM sweep pipeline: a band waited too long for its neighbour (results are invalid)");
        return MDISP_E_HIP;
    }
    return MDISP_OK;
}

static void* stage_buffer(mdisp_ctx* c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return c->stage;
    if (hipStreamSynchronize(c->stream) != hipSuccess) { mdisp_set_error("stream sync failed"); return nullptr; }
    if (c->stage) (void)hipFree(c->stage);
    c->stage = nullptr; c->stage_bytes = 0;
    const size_t want = md_align(bytes + bytes / 4, 1 << 20);
    if (hipMalloc(&c->stage, want) != hipSuccess) { mdisp_set_error("hipMalloc(%zu) failed for the copy staging buffer", want); return nullptr; }
    c->stage_bytes = want;
    return c->stage;
}

int md_h2d_2d(mdisp_ctx* c, void* d, size_t dpitch, const void* h, size_t hstep, size_t rowBytes, size_t rows)
{
    if (rows == 0 || rowBytes == 0) return MDISP_OK;
    if (hstep == dpitch) {
        MD_HIP(hipMemcpyAsync(d, h, (rows - 1) * dpitch + rowBytes, hipMemcpyHostToDevice, c->stream));
    } else if (hstep == rowBytes) {
        void* st = stage_buffer(c, rows * rowBytes);
        if (!st) return MDISP_E_NOMEM;
        MD_HIP(hipMemcpyAsync(st, h, rows * rowBytes, hipMemcpyHostToDevice, c->stream));
        MD_HIP(hipMemcpy2DAsync(d, dpitch, st, rowBytes, rowBytes, rows, hipMemcpyDeviceToDevice, c->stream));
    } else {
        MD_HIP(hipMemcpy2DAsync(d, dpitch, h, hstep, rowBytes, rows, hipMemcpyHostToDevice, c->stream));
    }
    return MDISP_OK;
}

int md_d2h_2d(mdisp_ctx* c, void* h, size_t hstep, const void* d, size_t dpitch, size_t rowBytes, size_t rows)
{
    if (rows == 0 || rowBytes == 0) return MDISP_OK;
    if (hstep == dpitch && hstep == rowBytes) {
        MD_HIP(hipMemcpyAsync(h, d, rows * rowBytes, hipMemcpyDeviceToHost, c->stream));
    } else if (hstep == rowBytes) {
        void* st = stage_buffer(c, rows * rowBytes);
        if (!st) return MDISP_E_NOMEM;
        MD_HIP(hipMemcpy2DAsync(st, rowBytes, d, dpitch, rowBytes, rows, hipMemcpyDeviceToDevice, c->stream));
        MD_HIP(hipMemcpyAsync(h, st, rows * rowBytes, hipMemcpyDeviceToHost, c->stream));
    } else {
        // the caller's rows are padded: never write between them
        MD_HIP(hipMemcpy2DAsync(h, hstep, d, dpitch, rowBytes, rows, hipMemcpyDeviceToHost, c->stream));
    }
    return MDISP_OK;
}

void* mdisp_ws(mdisp_ctx* c, size_t bytes)
{
    if (bytes <= c->ws_bytes) return c->ws;
    // the stream may still be reading the old block
    if (hipStreamSynchronize(c->stream) != hipSuccess) { mdisp_set_error("stream sync failed"); return nullptr; }
    if (c->ws) { (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
    size_t want = md_align(bytes + bytes / 8, 1 << 20);
    if (hipMalloc(&c->ws, want) != hipSuccess) {
        mdisp_set_error("hipMalloc(%zu) for the workspace failed", want);
        c->ws = nullptr;
        return nullptr;
    }
    c->ws_bytes = want;
    return c->ws;
}

extern "C" size_t mdisp_context_device_bytes(mdisp_ctx* c) { return c ? c->ws_bytes + c->stage_bytes : 0; }

extern "C" int mdisp_enable_timing(mdisp_ctx* c, int on)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    c->timing = on != 0;
    return MDISP_OK;
}

extern "C" int mdisp_last_kernel_ms(mdisp_ctx* c, int which, float* ms)
{
    MD_REQUIRE(c && ms && which >= 0 && which < 3, MDISP_E_BADARG, "bad arguments");
    MD_REQUIRE(c->ev_valid[which], MDISP_E_BADARG, "no timed call recorded (enable timing first)");
    MD_HIP(hipEventSynchronize(c->ev[2 * which + 1]));
    MD_HIP(hipEventElapsedTime(ms, c->ev[2 * which], c->ev[2 * which + 1]));
    return MDISP_OK;
}

// Streaming copy used to measure what HBM delivers on this device (the denominator next to the nominal peak): one
// 16-byte nontemporal load and store per lane and no loop, i.e. as many short-lived workgroups as the buffer has
// 4 KiB pieces.  Of the shapes scripts/copy_ubench.hip compares this one is the fastest (6.5 TB/s read + written on
// 2 GiB; grid-stride loops of 2048 .. 65536 workgroups reach 4.4 .. 5.9 TB/s, hipMemcpyAsync 4.9 TB/s).
typedef u32 md_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void md_stream_copy_kernel(const md_u32x4* __restrict__ src, md_u32x4* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int mdisp_copy_bandwidth_dev(mdisp_ctx* c, void* d_dst, const void* d_src, size_t bytes, int reps, float* gb_per_s)
{
    MD_REQUIRE(c && d_dst && d_src && gb_per_s, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(bytes >= 16 && bytes % 16 == 0 && reps > 0, MDISP_E_BADARG, "bytes must be a positive multiple of 16, reps > 0");
    MD_REQUIRE(((uintptr_t)d_dst | (uintptr_t)d_src) % 16 == 0, MDISP_E_BADARG, "buffers must be 16-byte aligned");
    hipEvent_t e0, e1;
    MD_HIP(hipEventCreate(&e0));
    MD_HIP(hipEventCreate(&e1));
    const size_t n = bytes / 16;
    MD_REQUIRE(n / 256 < 0x7FFFFFFFull, MDISP_E_BADARG, "buffer too large for one launch");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    md_stream_copy_kernel<<<blocks, 256, 0, c->stream>>>((const md_u32x4*)d_src, (md_u32x4*)d_dst, n);   // warm-up
    MD_HIP(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; r++) md_stream_copy_kernel<<<blocks, 256, 0, c->stream>>>((const md_u32x4*)d_src, (md_u32x4*)d_dst, n);
    MD_HIP(hipEventRecord(e1, c->stream));
    MD_HIP(hipGetLastError());
    MD_HIP(hipEventSynchronize(e1));
    float ms = 0;
    MD_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *gb_per_s = (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);
    return MDISP_OK;
}

// ---------------------------------------------------------------------------------------------
// helpers

static size_t depth_bytes(int depth)
{
    switch (depth) {
    case MDISP_8U: case MDISP_8S: return 1;
    case MDISP_16U: case MDISP_16S: return 2;
    case MDISP_32S: case MDISP_32F: return 4;
    case MDISP_64F: return 8;
    default: return 0;
    }
}

static int check_image(const mdisp_image* im, const char* what)
{
    MD_REQUIRE(im && im->data, MDISP_E_BADARG, "%s: empty image", what);
    MD_REQUIRE(im->rows > 0 && im->cols > 0 && im->channels > 0 && depth_bytes(im->depth) > 0, MDISP_E_BADARG,
               "%s: bad image header", what);
    MD_REQUIRE(im->step_bytes >= (size_t)im->cols * im->channels * depth_bytes(im->depth), MDISP_E_BADARG,
               "%s: step smaller than a row", what);
    return MDISP_OK;
}

static int check_out(const mdisp_image* im, int rows, int cols, int depth, int channels, const char* what)
{
    int rc = check_image(im, what);
    if (rc) return rc;
    MD_REQUIRE(im->rows == rows && im->cols == cols && im->depth == depth && im->channels == channels, MDISP_E_BADARG,
               "%s: expected %dx%d depth %d channels %d, got %dx%d depth %d channels %d", what, cols, rows, depth,
               channels, im->cols, im->rows, im->depth, im->channels);
    return MDISP_OK;
}

// bump allocator over the context workspace
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = md_align(off + bytes, 256); return o; }
};

// Upload an 8-bit image as single-channel gray (checkImg, methods_disp.cpp:130-147: 3-channel
// input goes through COLOR_BGR2GRAY).  `stage` is device scratch for the interleaved BGR rows.
static int upload_gray(mdisp_ctx* c, const mdisp_image* im, u8* d, size_t dpitch, u8* stage, size_t stage_pitch, int rgbOrder = 0)
{
    MD_REQUIRE(im->depth == MDISP_8U, MDISP_E_UNSUPPORTED, "only CV_8U images are supported (got depth %d)", im->depth);
    if (im->channels == 1) {
        MD_TRY(md_h2d_2d(c, d, dpitch, im->data, im->step_bytes, im->cols, im->rows));
        return MDISP_OK;
    }
    MD_REQUIRE(im->channels == 3, MDISP_E_UNSUPPORTED, "only 1- or 3-channel images are supported");
    MD_TRY(md_h2d_2d(c, stage, stage_pitch, im->data, im->step_bytes, (size_t)im->cols * 3, im->rows));
    return md_bgr2gray(c->stream, stage, stage_pitch, im->rows, im->cols, d, dpitch, rgbOrder);
}

static int census_arg_for(int method, int winSize, int leftBranch)
{
    // per-branch literals of censusStereo: methods_disp.cpp:885,927 vs :1028,1070
    if (method == MDISP_CIRCLE_CENSUS) return leftBranch ? 5 : winSize;
    if (method == MDISP_MULTISCALE_CENSUS) return leftBranch ? 16 : winSize;
    return winSize;
}

// ---------------------------------------------------------------------------------------------
// census / Hamming

extern "C" int mdisp_census_transform(mdisp_ctx* c, const mdisp_image* src, int method, int arg, mdisp_image* dst)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    if ((rc = check_out(dst, src->rows, src->cols, MDISP_8U, 1, "dst"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = src->rows, W = src->cols;
    const size_t pitch = md_align(W, 256), p3 = md_align((size_t)W * 3, 256);
    Carver cv;
    size_t o_src = cv.take(pitch * H), o_dst = cv.take(pitch * H), o_scr = cv.take(pitch * H + MD_CENSUS_SCRATCH_TAIL), o_stage = cv.take(p3 * H);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    if ((rc = upload_gray(c, src, base + o_src, pitch, base + o_stage, p3))) return rc;
    if ((rc = md_census(c->stream, base + o_src, pitch, H, W, method, arg, base + o_dst, pitch, base + o_scr))) return rc;
    MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, base + o_dst, pitch, W, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_hamming_dist(mdisp_ctx* c, const mdisp_image* a, const mdisp_image* b, mdisp_image* dst)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(a, "src1")) || (rc = check_image(b, "src2"))) return rc;
    MD_REQUIRE(a->rows == b->rows && a->cols == b->cols && a->depth == MDISP_8U && b->depth == MDISP_8U &&
               a->channels == 1 && b->channels == 1, MDISP_E_BADARG, "Hamming distance needs two CV_8UC1 images of equal size");
    if ((rc = check_out(dst, a->rows, a->cols, MDISP_32S, 1, "dst"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = a->rows, W = a->cols;
    const size_t pitch = md_align(W, 256), p4 = md_align((size_t)W * 4, 256);
    Carver cv;
    size_t o_a = cv.take(pitch * H), o_b = cv.take(pitch * H), o_d = cv.take(p4 * H);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_a, pitch, a->data, a->step_bytes, W, H));
    MD_TRY(md_h2d_2d(c, base + o_b, pitch, b->data, b->step_bytes, W, H));
    if ((rc = md_hamming(c->stream, base + o_a, pitch, base + o_b, pitch, H, W, (int32_t*)(base + o_d), p4))) return rc;
    MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, base + o_d, p4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

// ---------------------------------------------------------------------------------------------
// window-cost stereo (census stereo, SAD box, SAD integral)

static int check_pair(const mdisp_image* L, const mdisp_image* R, const mdisp_param* p)
{
    int rc;
    if ((rc = check_image(L, "imgLeft")) || (rc = check_image(R, "imgRight"))) return rc;
    MD_REQUIRE(p, MDISP_E_BADARG, "null parameters");
    // checkPairs + winSize parity, methods_disp.cpp:209-215
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG,
               "bad function parameters, please check image and window size.");
    MD_REQUIRE(p->winSize % 2 != 0 && p->winSize > 0, MDISP_E_BADARG,
               "bad function parameters, please check image and window size.");
    MD_REQUIRE(p->minDisparity >= 0 && p->maxDisparity >= p->minDisparity, MDISP_E_BADARG, "bad disparity range");
    return MDISP_OK;
}

struct PairBufs {
    u8 *ref, *other, *pad, *cRef, *cPad, *scr, *stage;
    size_t pitchW, pitchP, pitch3;
    void* out; int* I; int* best; void* extra;
};

// lay out and upload a pair; `ref` is the view the disparity is computed for
static int stage_pair(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p,
                      size_t out_bytes, bool need_sat, PairBufs* b, size_t extra_bytes = 0)
{
    const int H = L->rows, W = L->cols, Wp = W + p->maxDisparity;
    b->pitchW = md_align(W, 256); b->pitchP = md_align(Wp, 256); b->pitch3 = md_align((size_t)W * 3, 256);
    Carver cv;
    size_t o_ref = cv.take(b->pitchW * H), o_oth = cv.take(b->pitchW * H), o_pad = cv.take(b->pitchP * H);
    size_t o_cr = cv.take(b->pitchP * H), o_cp = cv.take(b->pitchP * H), o_scr = cv.take(b->pitchP * H + MD_CENSUS_SCRATCH_TAIL);
    size_t o_st = cv.take(b->pitch3 * H), o_out = cv.take(out_bytes);
    size_t o_I = 0, o_best = 0;
    if (need_sat) {
        const int nt = md_sad_inte_tables(H, W, p->maxDisparity - p->minDisparity + 1);
        o_I = cv.take((size_t)nt * (H + 1) * (W + 1) * 4); o_best = cv.take((size_t)H * W * 8);
    }
    size_t o_extra = cv.take(extra_bytes);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    b->ref = base + o_ref; b->other = base + o_oth; b->pad = base + o_pad; b->cRef = base + o_cr; b->cPad = base + o_cp;
    b->scr = base + o_scr; b->stage = base + o_st; b->out = base + o_out;
    b->I = (int*)(base + o_I); b->best = (int*)(base + o_best);
    b->extra = base + o_extra;
    int rc;
    const mdisp_image* refIm = p->isDispLeft ? L : R;
    const mdisp_image* othIm = p->isDispLeft ? R : L;
    if ((rc = upload_gray(c, refIm, b->ref, b->pitchW, b->stage, b->pitch3))) return rc;
    if ((rc = upload_gray(c, othIm, b->other, b->pitchW, b->stage, b->pitch3))) return rc;
    return MDISP_OK;
}

// Device-side body shared by the host and _dev census-stereo entry points.
static int census_stereo_body(mdisp_ctx* c, const u8* ref, const u8* other, size_t pitch, int H, int W,
                              const mdisp_param* p, int method, int agg, u8* pad, size_t pitchP, u8* cRef, size_t pitchC,
                              u8* cPad, u8* scr, double* dOut, void* gifScratch)
{
    const int maxD = p->maxDisparity, minD = p->minDisparity, Wp = W + maxD;
    int rc;
    hipStream_t s = c->stream;
    MD_REQUIRE(agg >= MDISP_AGG_NONE && agg <= MDISP_AGG_GUIDED, MDISP_E_BADARG, "unknown aggregation mode %d", agg);
    // left-disparity branch pads the right image on the left (:862), right-disparity branch pads
    // the left image on the right (:1005); both BORDER_REFLECT
    if (p->isDispLeft) rc = md_pad_reflect_lr(s, other, pitch, H, W, maxD, 0, pad, pitchP);
    else rc = md_pad_reflect_lr(s, other, pitch, H, W, 0, maxD, pad, pitchP);
    if (rc) return rc;
    const int arg = census_arg_for(method, p->winSize, p->isDispLeft);
    if ((rc = md_census(s, ref, pitch, H, W, method, arg, cRef, pitchC, scr))) return rc;
    if ((rc = md_census(s, pad, pitchP, H, Wp, method, arg, cPad, pitchP, scr))) return rc;
    const int off0 = p->isDispLeft ? (maxD - minD) : minD, offStep = p->isDispLeft ? -1 : 1;
    if (agg == MDISP_AGG_GUIDED) {
        // getGuidedFilter(img, slice, winSize, 1e-6) per slice (:978, :1120).  The left-disparity
        // branch narrows the slice to CV_8U first (normalised in float), the right-disparity
        // branch passes CV_32S (normalised in double).
        return md_gif_stereo(c, gifScratch, ref, pitch, H, W, maxD - minD + 1, minD, p->winSize, 1e-6, MDISP_GIF, 0, 0.0,
                             cRef, pitchC, cPad, pitchP, off0, offStep, MD_COST_HAMMING, nullptr, p->isDispLeft ? 0 : 1,
                             nullptr, dOut);
    }
    const int win = agg == MDISP_AGG_NONE ? 1 : p->winSize;
    const int kind = agg == MDISP_AGG_SQRBOX ? MD_COST_HAMMING_SQR : MD_COST_HAMMING;
    return md_cost_box_wta(s, cRef, pitchC, cPad, pitchP, H, W, minD, maxD, off0, offStep, win, kind, MD_OUT_F64, dOut, W);
}

extern "C" int mdisp_census_stereo(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p,
                                   int census_method, int agg_mode, mdisp_image* out)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_pair(L, R, p))) return rc;
    if ((rc = check_out(out, L->rows, L->cols, MDISP_64F, 1, "out"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols;
    PairBufs b;
    const size_t extra = agg_mode == MDISP_AGG_GUIDED ? md_gif_scratch_bytes(H, W, p->maxDisparity - p->minDisparity + 1) : 0;
    if ((rc = stage_pair(c, L, R, p, (size_t)H * W * 8, false, &b, extra))) return rc;
    // census rows are read as dwords: the census of `ref` uses the padded-width pitch as well
    if ((rc = census_stereo_body(c, b.ref, b.other, b.pitchW, H, W, p, census_method, agg_mode, b.pad, b.pitchP, b.cRef,
                                 b.pitchP, b.cPad, b.scr, (double*)b.out, b.extra))) return rc;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, b.out, (size_t)W * 8, (size_t)W * 8, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_census_stereo_dev(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                                       size_t pitch, const mdisp_param* p, int census_method, int agg_mode, double* d_disp)
{
    MD_REQUIRE(c && d_left && d_right && d_disp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(rows > 0 && cols > 0 && pitch >= (size_t)cols && pitch % 4 == 0, MDISP_E_BADARG,
               "bad geometry (pitch must be a multiple of 4 and >= cols)");
    MD_REQUIRE(p->winSize % 2 != 0 && p->winSize > 0 && p->minDisparity >= 0 && p->maxDisparity >= p->minDisparity,
               MDISP_E_BADARG, "bad function parameters, please check image and window size.");
    MD_HIP(hipSetDevice(c->device));
    const int Wp = cols + p->maxDisparity;
    const size_t pitchP = md_align(Wp, 256);
    Carver cv;
    size_t o_pad = cv.take(pitchP * rows), o_cr = cv.take(pitchP * rows), o_cp = cv.take(pitchP * rows), o_scr = cv.take(pitchP * rows + MD_CENSUS_SCRATCH_TAIL);
    size_t o_gif = cv.take(agg_mode == MDISP_AGG_GUIDED ? md_gif_scratch_bytes(rows, cols, p->maxDisparity - p->minDisparity + 1) : 0);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    const u8* ref = p->isDispLeft ? d_left : d_right;
    const u8* other = p->isDispLeft ? d_right : d_left;
    return census_stereo_body(c, ref, other, pitch, rows, cols, p, census_method, agg_mode, base + o_pad, pitchP,
                              base + o_cr, pitchP, base + o_cp, base + o_scr, d_disp, base + o_gif);
}

// The same call without any allocation: the three padded-width planes and the census scratch live in the library's
// file-scope device array instead of the context's grow-only workspace.
extern "C" int mdisp_census_stereo_dev_noalloc(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols,
                                               size_t pitch, const mdisp_param* p, int census_method, int agg_mode, double* d_disp)
{
    MD_REQUIRE(c && d_left && d_right && d_disp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(rows > 0 && cols > 0 && pitch >= (size_t)cols && pitch % 4 == 0, MDISP_E_BADARG,
               "bad geometry (pitch must be a multiple of 4 and >= cols)");
    MD_REQUIRE(p->winSize % 2 != 0 && p->winSize > 0 && p->minDisparity >= 0 && p->maxDisparity >= p->minDisparity,
               MDISP_E_BADARG, "bad function parameters, please check image and window size.");
    MD_REQUIRE(agg_mode != MDISP_AGG_GUIDED, MDISP_E_UNSUPPORTED,
               "guided aggregation needs per-slice planes far beyond the fixed scratch: use mdisp_census_stereo_dev");
    const size_t pitchP = md_align((size_t)cols + p->maxDisparity, 256);
    Carver cv;
    size_t o_pad = cv.take(pitchP * rows), o_cr = cv.take(pitchP * rows), o_cp = cv.take(pitchP * rows), o_scr = cv.take(pitchP * rows + MD_CENSUS_SCRATCH_TAIL);
    MD_REQUIRE(cv.off <= (size_t)MDISP_NOALLOC_SCRATCH_BYTES, MDISP_E_RANGE,
               "%dx%d with maxDisparity %d needs %zu bytes of scratch, the fixed array holds %u", cols, rows, p->maxDisparity, cv.off,
               (unsigned)MDISP_NOALLOC_SCRATCH_BYTES);
    MD_HIP(hipSetDevice(c->device));
    if (!c->noalloc_scratch) MD_TRY(scratch_of_device(c->device, &c->noalloc_scratch));
    u8* base = c->noalloc_scratch;     // the device address of `scratch` on the context's GPU, never the symbol itself
    const u8* ref = p->isDispLeft ? d_left : d_right;
    const u8* other = p->isDispLeft ? d_right : d_left;
    return census_stereo_body(c, ref, other, pitch, rows, cols, p, census_method, agg_mode, base + o_pad, pitchP,
                              base + o_cr, pitchP, base + o_cp, base + o_scr, d_disp, nullptr);
}

static int sad_common(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, mdisp_image* out, bool inte,
                      int tableMode = MDISP_SAT_F32)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_pair(L, R, p))) return rc;
    if ((rc = check_out(out, L->rows, L->cols, MDISP_8U, 1, "out"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols, maxD = p->maxDisparity, minD = p->minDisparity;
    PairBufs b;
    const size_t opitch = md_align(W, 256);
    if ((rc = stage_pair(c, L, R, p, opitch * H, inte, &b))) return rc;
    hipStream_t s = c->stream;
    if (p->isDispLeft) rc = md_pad_reflect_lr(s, b.other, b.pitchW, H, W, maxD, 0, b.pad, b.pitchP);
    else rc = md_pad_reflect_lr(s, b.other, b.pitchW, H, W, 0, maxD, b.pad, b.pitchP);
    if (rc) return rc;
    const int off0 = p->isDispLeft ? (maxD - minD) : minD, offStep = p->isDispLeft ? -1 : 1;
    if (inte) rc = md_sad_inte(s, b.ref, b.pitchW, b.pad, b.pitchP, H, W, minD, maxD, off0, offStep, p->winSize,
                               tableMode == MDISP_SAT_F32, (u8*)b.out, opitch, b.I, b.best);
    else rc = md_cost_box_wta(s, b.ref, b.pitchW, b.pad, b.pitchP, H, W, minD, maxD, off0, offStep, p->winSize,
                              MD_COST_ABSDIFF_SQR, MD_OUT_U8, b.out, opitch);
    if (rc) return rc;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, b.out, opitch, W, H));
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}

extern "C" int mdisp_sad_inte(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, mdisp_image* out)
{ return sad_common(c, L, R, p, out, true, md_tune(MDISP_TUNE_SAT_INT_TABLES) > 0 ? MDISP_SAT_INT : MDISP_SAT_F32); }

extern "C" int mdisp_sad_inte_mode(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, int table_mode,
                                   mdisp_image* out)
{
    MD_REQUIRE(table_mode == MDISP_SAT_F32 || table_mode == MDISP_SAT_INT, MDISP_E_BADARG, "unknown table mode %d", table_mode);
    return sad_common(c, L, R, p, out, true, table_mode);
}

extern "C" int mdisp_sad_box(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, mdisp_image* out)
{ return sad_common(c, L, R, p, out, false); }

extern "C" int mdisp_asw_gifs(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, double eps,
                              int gif_type, int r2, double namda, double h, int cost_source, const float* costs, mdisp_image* out)
{
    (void)h;   // accepted and unused, as in the reference's filters
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_pair(L, R, p))) return rc;
    if ((rc = check_out(out, L->rows, L->cols, MDISP_32F, 1, "out"))) return rc;
    MD_REQUIRE(gif_type >= MDISP_GIF && gif_type <= MDISP_OURS_GIF3, MDISP_E_BADARG, "unknown GIF type %d", gif_type);
    MD_REQUIRE(cost_source != MDISP_COST_VOLUME || costs, MDISP_E_BADARG, "cost volume source needs the slices");
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols, maxD = p->maxDisparity, minD = p->minDisparity, D = maxD - minD + 1, Wp = W + maxD;
    PairBufs b;
    if ((rc = stage_pair(c, L, R, p, (size_t)H * W * 4, false, &b, md_gif_scratch_bytes(H, W, D)))) return rc;
    hipStream_t s = c->stream;
    const int off0 = p->isDispLeft ? (maxD - minD) : minD, offStep = p->isDispLeft ? -1 : 1;
    const u8 *cr = nullptr, *cp = nullptr; size_t crp = 0, cpp = 0; int kind = -1;
    if (cost_source != MDISP_COST_VOLUME) {
        if (p->isDispLeft) rc = md_pad_reflect_lr(s, b.other, b.pitchW, H, W, maxD, 0, b.pad, b.pitchP);
        else rc = md_pad_reflect_lr(s, b.other, b.pitchW, H, W, 0, maxD, b.pad, b.pitchP);
        if (rc) return rc;
        if (cost_source == MDISP_COST_CENSUS) {
            if ((rc = md_census(s, b.ref, b.pitchW, H, W, MDISP_BASIC_CENSUS, 0, b.cRef, b.pitchP, b.scr))) return rc;
            if ((rc = md_census(s, b.pad, b.pitchP, H, Wp, MDISP_BASIC_CENSUS, 0, b.cPad, b.pitchP, b.scr))) return rc;
            cr = b.cRef; crp = b.pitchP; cp = b.cPad; cpp = b.pitchP; kind = MD_COST_HAMMING;
        } else if (cost_source == MDISP_COST_ABSDIFF) {
            cr = b.ref; crp = b.pitchW; cp = b.pad; cpp = b.pitchP; kind = MD_COST_ABSDIFF;
        } else { mdisp_set_error("unknown cost source %d", cost_source); return MDISP_E_BADARG; }
    }
    if ((rc = md_gif_stereo(c, b.extra, b.ref, b.pitchW, H, W, D, minD, p->winSize, eps, gif_type, r2, namda, cr, crp, cp, cpp,
                            off0, offStep, kind, cost_source == MDISP_COST_VOLUME ? costs : nullptr, 0, (float*)b.out, nullptr))) return rc;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, b.out, (size_t)W * 4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}

// Device-resident asw_gifs restricted to the output rows [row_begin, row_end): what one GPU computes
// when a large pair is sharded by row bands.  Both full images are resident on every GPU; everything
// that is global to the image (guide statistics, steer weights, per-slice cost ranges) is recomputed
// locally from them, so the bands need no exchange and are bit-identical to the rows of a full call.
struct BandSetup { const u8 *ref, *cr, *cp; size_t crp, cpp; int kind, off0, offStep, H, W, D, minD; void* gif; };
// checks + the full-image preliminaries every band needs (padded other view, census codes of both views)
static int band_setup(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch, const mdisp_param* p,
                      int gif_type, int cost_source, int row_begin, int row_end, BandSetup* b)
{
    MD_REQUIRE(c && d_left && d_right && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(rows > 0 && cols > 0 && pitch >= (size_t)cols && pitch % 4 == 0, MDISP_E_BADARG,
               "bad geometry (pitch must be a multiple of 4 and >= cols)");
    MD_REQUIRE(p->winSize % 2 != 0 && p->winSize > 0 && p->minDisparity >= 0 && p->maxDisparity >= p->minDisparity,
               MDISP_E_BADARG, "bad function parameters, please check image and window size.");
    MD_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= rows, MDISP_E_BADARG, "bad row band [%d, %d)", row_begin, row_end);
    MD_REQUIRE(gif_type >= MDISP_GIF && gif_type <= MDISP_OURS_GIF3, MDISP_E_BADARG, "unknown GIF type %d", gif_type);
    MD_REQUIRE(cost_source == MDISP_COST_CENSUS || cost_source == MDISP_COST_ABSDIFF, MDISP_E_BADARG,
               "the band entry points form their cost slices on the device (census or absdiff)");
    MD_HIP(hipSetDevice(c->device));
    const int H = rows, W = cols, maxD = p->maxDisparity, minD = p->minDisparity, D = maxD - minD + 1, Wp = W + maxD;
    const size_t pitchP = md_align(Wp, 256);
    Carver cv;
    size_t o_pad = cv.take(pitchP * H), o_cr = cv.take(pitchP * H), o_cp = cv.take(pitchP * H);
    size_t o_scr = cv.take(pitchP * H + MD_CENSUS_SCRATCH_TAIL), o_gif = cv.take(md_gif_scratch_bytes(H, W, D));
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    hipStream_t s = c->stream;
    const u8* ref = p->isDispLeft ? d_left : d_right;
    const u8* other = p->isDispLeft ? d_right : d_left;
    int rc;
    if (p->isDispLeft) rc = md_pad_reflect_lr(s, other, pitch, H, W, maxD, 0, base + o_pad, pitchP);
    else rc = md_pad_reflect_lr(s, other, pitch, H, W, 0, maxD, base + o_pad, pitchP);
    if (rc) return rc;
    b->off0 = p->isDispLeft ? (maxD - minD) : minD; b->offStep = p->isDispLeft ? -1 : 1;
    if (cost_source == MDISP_COST_CENSUS) {
        if ((rc = md_census(s, ref, pitch, H, W, MDISP_BASIC_CENSUS, 0, base + o_cr, pitchP, base + o_scr))) return rc;
        if ((rc = md_census(s, base + o_pad, pitchP, H, Wp, MDISP_BASIC_CENSUS, 0, base + o_cp, pitchP, base + o_scr))) return rc;
        b->cr = base + o_cr; b->crp = pitchP; b->cp = base + o_cp; b->cpp = pitchP; b->kind = MD_COST_HAMMING;
    } else {
        b->cr = ref; b->crp = pitch; b->cp = base + o_pad; b->cpp = pitchP; b->kind = MD_COST_ABSDIFF;
    }
    b->ref = ref; b->H = H; b->W = W; b->D = D; b->minD = minD; b->gif = base + o_gif;
    return MDISP_OK;
}

extern "C" int mdisp_asw_gifs_band_dev(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                                       const mdisp_param* p, double eps, int gif_type, int r2, double namda, int cost_source,
                                       int row_begin, int row_end, float* d_disp, const float* cost_minmax, const double* steer_max)
{
    MD_REQUIRE(d_disp, MDISP_E_BADARG, "null argument");
    BandSetup b;
    MD_TRY(band_setup(c, d_left, d_right, rows, cols, pitch, p, gif_type, cost_source, row_begin, row_end, &b));
    if (row_begin == row_end) return MDISP_OK;
    return md_gif_stereo(c, b.gif, b.ref, pitch, b.H, b.W, b.D, b.minD, p->winSize, eps, gif_type, r2, namda, b.cr, b.crp, b.cp, b.cpp,
                         b.off0, b.offStep, b.kind, nullptr, 0, d_disp, nullptr, row_begin, row_end, cost_minmax,
                         gif_type == MDISP_SKWGIF ? steer_max : nullptr);
}

extern "C" int mdisp_asw_gifs_band_stats_dev(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                                             const mdisp_param* p, int gif_type, int cost_source, int row_begin, int row_end,
                                             float* cost_minmax, double* steer_max)
{
    MD_REQUIRE(cost_minmax && steer_max, MDISP_E_BADARG, "null argument");
    BandSetup b;
    MD_TRY(band_setup(c, d_left, d_right, rows, cols, pitch, p, gif_type, cost_source, row_begin, row_end, &b));
    return md_gif_band_stats(c, b.gif, b.ref, pitch, b.H, b.W, b.D, gif_type, b.cr, b.crp, b.cp, b.cpp, b.off0, b.offStep, b.kind,
                             row_begin, row_end, cost_minmax, steer_max);
}

// ---------------------------------------------------------------------------------------------
// disparity -> 3-D points

static int reproject_common(mdisp_ctx* c, const mdisp_image* disp, const mdisp_image* bgr, mdisp_image* xyz, const double* Q,
                            const double* K, double baseline)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(disp, "disparity"))) return rc;
    MD_REQUIRE(disp->depth == MDISP_32F && disp->channels == 1, MDISP_E_BADARG, "the disparity map must be CV_32FC1");
    if ((rc = check_out(xyz, disp->rows, disp->cols, MDISP_32F, 3, "xyz"))) return rc;
    if (bgr) {
        if ((rc = check_image(bgr, "colour image"))) return rc;
        MD_REQUIRE(bgr->depth == MDISP_32F && bgr->channels == 3 && bgr->rows == disp->rows && bgr->cols == disp->cols, MDISP_E_BADARG,
                   "the colour image must be CV_32FC3 of the disparity's size");
    }
    MD_HIP(hipSetDevice(c->device));
    const int H = disp->rows, W = disp->cols;
    Carver cv;
    size_t o_d = cv.take((size_t)H * W * 4), o_c = cv.take(bgr ? (size_t)H * W * 12 : 0), o_x = cv.take((size_t)H * W * 12);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_d, (size_t)W * 4, disp->data, disp->step_bytes, (size_t)W * 4, H));
    if (bgr) MD_TRY(md_h2d_2d(c, base + o_c, (size_t)W * 12, bgr->data, bgr->step_bytes, (size_t)W * 12, H));
    if (Q) rc = md_reproject_q(c->stream, (const float*)(base + o_d), H, W, Q, (float*)(base + o_x));
    else rc = md_reproject_tri(c->stream, (const float*)(base + o_d), H, W, K, baseline, bgr ? (const float*)(base + o_c) : nullptr,
                               (float*)(base + o_x));
    if (rc) return rc;
    MD_TRY(md_d2h_2d(c, xyz->data, xyz->step_bytes, base + o_x, (size_t)W * 12, (size_t)W * 12, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_reproject_q(mdisp_ctx* c, const mdisp_image* disp, const double Q[16], mdisp_image* xyz)
{
    MD_REQUIRE(Q, MDISP_E_BADARG, "null Q");
    return reproject_common(c, disp, nullptr, xyz, Q, nullptr, 0.0);
}

extern "C" int mdisp_reproject_triangulate(mdisp_ctx* c, const mdisp_image* disp, const double K[9], double baseline,
                                           const mdisp_image* bgr, mdisp_image* xyz)
{
    MD_REQUIRE(K, MDISP_E_BADARG, "null camera matrix");
    return reproject_common(c, disp, bgr, xyz, nullptr, K, baseline);
}

// ---------------------------------------------------------------------------------------------
// SGM

extern "C" size_t mdisp_sgbm_workspace_bytes(int rows, int cols, const mdisp_sgbm_params* p)
{
    md_sgm_plan pl;
    if (md_sgm_plan_make(rows, cols, p, 1, &pl)) return 0;
    return pl.total;
}

extern "C" size_t mdisp_sgbm_batch_workspace_bytes(int rows, int cols, int batch, const mdisp_sgbm_params* p)
{
    md_sgm_plan pl;
    if (md_sgm_plan_make(rows, cols, p, batch, &pl)) return 0;
    return pl.total;
}

extern "C" int mdisp_sgbm_batch_dev(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int batch, size_t pair_stride,
                                    int rows, int cols, size_t pitch, const mdisp_sgbm_params* p, int16_t* d_disp,
                                    void* d_workspace, size_t workspace_bytes)
{
    MD_REQUIRE(c && d_left && d_right && d_disp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(pitch >= (size_t)cols && (batch == 1 || pair_stride >= pitch * rows), MDISP_E_BADARG, "bad pitch / pair stride");
    md_sgm_plan pl;
    int rc;
    if ((rc = md_sgm_plan_make(rows, cols, p, batch, &pl))) return rc;
    MD_HIP(hipSetDevice(c->device));
    void* ws = d_workspace;
    if (ws) MD_REQUIRE(workspace_bytes >= pl.total, MDISP_E_BADARG, "workspace too small: %zu < %zu", workspace_bytes, pl.total);
    else { ws = mdisp_ws(c, pl.total); if (!ws) return MDISP_E_NOMEM; }
    if (c->timing) MD_HIP(hipEventRecord(c->ev[0], c->stream));
    if ((rc = md_sgm_run(c, pl, d_left, d_right, pitch, pair_stride, d_disp, ws))) return rc;
    if (c->timing) { MD_HIP(hipEventRecord(c->ev[1], c->stream)); c->ev_valid[0] = true; }
    return MDISP_OK;
}

extern "C" int mdisp_sgbm_rows_geometry(int rows, int cols, const mdisp_sgbm_params* p, int* band_rows, size_t* boundary_bytes, size_t* workspace_bytes)
{
    md_sgm_plan pl;
    MD_TRY(md_sgm_plan_make(rows, cols, p, 1, &pl, false, true));
    if (band_rows) *band_rows = md_sgm2_band_rows(pl);
    if (boundary_bytes) *boundary_bytes = md_sgm2_boundary_bytes(pl);
    if (workspace_bytes) *workspace_bytes = pl.total;
    return MDISP_OK;
}

extern "C" int mdisp_sgbm_rows_sweep_dev(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                                         const mdisp_sgbm_params* p, int row_begin, int row_end, int direction, int second,
                                         const void* d_boundary_in, void* d_boundary_out, int16_t* d_disp, void* d_workspace, size_t workspace_bytes)
{
    MD_REQUIRE(c && d_left && d_right && p && d_workspace, MDISP_E_BADARG, "null argument (the workspace carries state from the first sweep to the second: the caller must hold it)");
    MD_REQUIRE(pitch >= (size_t)cols, MDISP_E_BADARG, "pitch smaller than a row");
    MD_REQUIRE((direction == 0 || direction == 1) && (second == 0 || second == 1), MDISP_E_BADARG, "direction and second are 0 or 1");
    md_sgm_plan pl;
    MD_TRY(md_sgm_plan_make(rows, cols, p, 1, &pl, false, true));
    MD_REQUIRE(workspace_bytes >= pl.total, MDISP_E_BADARG, "workspace too small: %zu < %zu", workspace_bytes, pl.total);
    MD_HIP(hipSetDevice(c->device));
    return md_sgm_rows_sweep(c, pl, d_left, d_right, pitch, d_workspace, row_begin, row_end, direction == 1, second == 1,
                             (const u8*)d_boundary_in, (u8*)d_boundary_out, d_disp);
}

extern "C" int mdisp_sgbm_dev(mdisp_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                              const mdisp_sgbm_params* p, int16_t* d_disp, void* d_workspace, size_t workspace_bytes)
{
    MD_REQUIRE(c && d_left && d_right && d_disp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(pitch >= (size_t)cols, MDISP_E_BADARG, "pitch smaller than a row");
    return mdisp_sgbm_batch_dev(c, d_left, d_right, 1, 0, rows, cols, pitch, p, d_disp, d_workspace, workspace_bytes);
}

extern "C" int mdisp_sgbm(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_sgbm_params* p, mdisp_image* out)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(L, "left")) || (rc = check_image(R, "right"))) return rc;
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG, "left/right size mismatch");
    if ((rc = check_out(out, L->rows, L->cols, MDISP_16S, 1, "out"))) return rc;
    md_sgm_plan pl;
    if ((rc = md_sgm_plan_make(L->rows, L->cols, p, 1, &pl))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols;
    const size_t pitch = md_align(W, 256), p3 = md_align((size_t)W * 3, 256);
    Carver cv;
    size_t o_ws = cv.take(pl.total), o_l = cv.take(pitch * H), o_r = cv.take(pitch * H), o_st = cv.take(p3 * H), o_d = cv.take((size_t)H * W * 2);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    if ((rc = upload_gray(c, L, base + o_l, pitch, base + o_st, p3))) return rc;
    if ((rc = upload_gray(c, R, base + o_r, pitch, base + o_st, p3))) return rc;
    if ((rc = md_sgm_run(c, pl, base + o_l, base + o_r, pitch, 0, (int16_t*)(base + o_d), base + o_ws))) return rc;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 2, (size_t)W * 2, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return md_check_sgm_error(c);
}

extern "C" int mdisp_legacy_disparity(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, int method, int winSize, int T,
                                      int minD, int numD, mdisp_image* out)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(L, "left")) || (rc = check_image(R, "right"))) return rc;
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG, "left/right size mismatch");
    if ((rc = check_out(out, L->rows, L->cols, MDISP_8U, 1, "out"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols;
    const size_t pitch = md_align(W, 256), p3 = md_align((size_t)W * 3, 256);
    Carver cv;
    size_t o_l = cv.take(pitch * H), o_r = cv.take(pitch * H), o_st = cv.take(p3 * H), o_d = cv.take(pitch * H);
    size_t o_I = 0, o_best = 0;
    const bool tables = md_legacy_uses_tables(method) && numD > 0;
    if (tables) { o_I = cv.take((size_t)md_sad_inte_tables(H, W, numD) * (H + 1) * (W + 1) * 4); o_best = cv.take((size_t)H * W * 8); }
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    if ((rc = upload_gray(c, L, base + o_l, pitch, base + o_st, p3))) return rc;
    if ((rc = upload_gray(c, R, base + o_r, pitch, base + o_st, p3))) return rc;
    if ((rc = md_legacy_disparity(c->stream, base + o_l, pitch, base + o_r, pitch, H, W, method, winSize, T, minD, numD, base + o_d, pitch,
                                  tables ? (int*)(base + o_I) : nullptr, tables ? (int*)(base + o_best) : nullptr)))
        return rc;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, pitch, W, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

// stereomatch_1::computeNCC: the map (out, CV_32F) and / or the normalised cost volume (cost, numDisparity slices of rows*cols floats)
extern "C" int mdisp_ncc(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, int is_left, int winSize, int minD, int numD,
                         mdisp_image* out, float* cost)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(L, "left")) || (rc = check_image(R, "right"))) return rc;
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG, "left/right size mismatch");
    MD_REQUIRE(out || cost, MDISP_E_BADARG, "nothing to compute: pass a map, a cost volume or both");
    MD_REQUIRE(winSize >= 1 && winSize % 2 == 1, MDISP_E_BADARG, "winSize must be odd");
    MD_REQUIRE(numD >= 1 && minD >= 0 && (long)L->cols + minD + numD <= 65535, MDISP_E_BADARG, "disparity range");
    if (out && (rc = check_out(out, L->rows, L->cols, MDISP_32F, 1, "out"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols;
    const size_t pitch = md_align(W, 256), p3 = md_align((size_t)W * 3, 256), N = (size_t)H * W;
    Carver cv;
    const size_t o_l = cv.take(pitch * H), o_r = cv.take(pitch * H), o_st = cv.take(p3 * H), o_ws = cv.take(md_ncc_bytes(H, W, minD, numD)),
                 o_d = cv.take(N * 4), o_c = cv.take(cost ? N * 4 * numD : 0);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    // the reference converts colour input with COLOR_RGB2GRAY (method_StereoMatching.cpp:934-942): the first channel is weighted as red
    if ((rc = upload_gray(c, L, base + o_l, pitch, base + o_st, p3, 1))) return rc;
    if ((rc = upload_gray(c, R, base + o_r, pitch, base + o_st, p3, 1))) return rc;
    MD_TRY(md_ncc(c->stream, base + o_l, base + o_r, pitch, H, W, is_left != 0, winSize, minD, numD, base + o_ws,
                  out ? (float*)(base + o_d) : nullptr, cost ? (float*)(base + o_c) : nullptr));
    if (out) MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 4, (size_t)W * 4, H));
    if (cost) MD_HIP(hipMemcpyAsync(cost, base + o_c, N * 4 * numD, hipMemcpyDeviceToHost, c->stream));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

// A sequence of pairs in host memory, processed in chunks: the copies of chunk k+1 and of the results of
// chunk k-1 travel on a second stream while chunk k is being computed.
extern "C" int mdisp_sgbm_batch(mdisp_ctx* c, const uint8_t* const* lefts, const uint8_t* const* rights, int npairs, int rows, int cols,
                                size_t step_bytes, const mdisp_sgbm_params* p, int16_t* const* disps, int chunk)
{
    MD_REQUIRE(c && lefts && rights && disps && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(npairs >= 0 && rows > 0 && cols > 0 && step_bytes >= (size_t)cols, MDISP_E_BADARG, "bad geometry");
    if (npairs == 0) return MDISP_OK;
    for (int i = 0; i < npairs; i++) MD_REQUIRE(lefts[i] && rights[i] && disps[i], MDISP_E_BADARG, "pair %d: null image", i);
    if (chunk <= 0) chunk = 16;
    if (chunk > npairs) chunk = npairs;
    md_sgm_plan pl, plTail;
    MD_TRY(md_sgm_plan_make(rows, cols, p, chunk, &pl));
    const int nchunks = (npairs + chunk - 1) / chunk, tail = npairs - (nchunks - 1) * chunk;
    MD_TRY(md_sgm_plan_make(rows, cols, p, tail, &plTail));
    MD_HIP(hipSetDevice(c->device));
    const int H = rows, W = cols;
    const size_t pitch = md_align((size_t)W, 4), img = pitch * H, out = (size_t)H * W * 2;
    Carver cv;
    const size_t o_ws = cv.take(pl.total > plTail.total ? pl.total : plTail.total);
    size_t o_in[2], o_out[2];
    for (int s = 0; s < 2; s++) { o_in[s] = cv.take(2 * img * chunk); o_out[s] = cv.take(out * chunk); }
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    hipStream_t copy = nullptr;
    hipEvent_t ev[6] = {};      // per slot: inputs uploaded, chunk computed, results downloaded
    int rc = MDISP_OK;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        mdisp_set_error("mdisp_sgbm_batch: %s -> %s", what, hipGetErrorString(e));
        rc = MDISP_E_HIP;
        return false;
    };
    if (!hip_ok(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking), "hipStreamCreate")) return rc;
    for (int i = 0; i < 6 && rc == MDISP_OK; i++) hip_ok(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), "hipEventCreate");
    // the copy stream starts after whatever the context's stream still has in flight on this workspace
    if (rc == MDISP_OK && hip_ok(hipEventRecord(ev[0], c->stream), "hipEventRecord")) hip_ok(hipStreamWaitEvent(copy, ev[0], 0), "hipStreamWaitEvent");
    auto download = [&](int k) {          // results of chunk k
        const int slot = k & 1, first = k * chunk, n = k == nchunks - 1 ? tail : chunk;
        if (!hip_ok(hipStreamWaitEvent(copy, ev[2 + slot], 0), "hipStreamWaitEvent")) return;
        for (int i = 0; i < n && rc == MDISP_OK; i++)
            hip_ok(hipMemcpyAsync(disps[first + i], base + o_out[slot] + (size_t)i * out, out, hipMemcpyDeviceToHost, copy), "download");
        if (rc == MDISP_OK) hip_ok(hipEventRecord(ev[4 + slot], copy), "hipEventRecord");
    };
    for (int k = 0; k < nchunks && rc == MDISP_OK; k++) {
        const int slot = k & 1, first = k * chunk, n = k == nchunks - 1 ? tail : chunk;
        if (k >= 2 && !hip_ok(hipStreamWaitEvent(copy, ev[2 + slot], 0), "hipStreamWaitEvent")) break;   // chunk k-2 has read this slot
        u8* dIn = base + o_in[slot];
        for (int i = 0; i < n && rc == MDISP_OK; i++) {
            for (int side = 0; side < 2 && rc == MDISP_OK; side++) {
                const uint8_t* src = side ? rights[first + i] : lefts[first + i];
                u8* dst = dIn + ((size_t)side * chunk + i) * img;
                if (step_bytes == pitch) hip_ok(hipMemcpyAsync(dst, src, img - (pitch - W), hipMemcpyHostToDevice, copy), "upload");
                else hip_ok(hipMemcpy2DAsync(dst, pitch, src, step_bytes, W, H, hipMemcpyHostToDevice, copy), "upload");
            }
        }
        if (rc != MDISP_OK || !hip_ok(hipEventRecord(ev[slot], copy), "hipEventRecord")) break;
        if (!hip_ok(hipStreamWaitEvent(c->stream, ev[slot], 0), "hipStreamWaitEvent")) break;
        if (k >= 2 && !hip_ok(hipStreamWaitEvent(c->stream, ev[4 + slot], 0), "hipStreamWaitEvent")) break;   // results of chunk k-2 are out
        rc = md_sgm_run(c, k == nchunks - 1 ? plTail : pl, dIn, dIn + (size_t)chunk * img, pitch, img, (int16_t*)(base + o_out[slot]), base + o_ws);
        if (rc != MDISP_OK || !hip_ok(hipEventRecord(ev[2 + slot], c->stream), "hipEventRecord")) break;
        if (k >= 1) download(k - 1);
    }
    if (rc == MDISP_OK) download(nchunks - 1);
    // drain both streams whatever happened, then release the per-call objects
    (void)hipStreamSynchronize(copy);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 6; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
    (void)hipStreamDestroy(copy);
    if (rc != MDISP_OK) return rc;
    return md_check_sgm_error(c);
}

// Upload an 8-bit 1- or 3-channel image as it is (interleaved), rows at `dpitch`
static int upload_u8(mdisp_ctx* c, const mdisp_image* im, u8* d, size_t dpitch)
{
    MD_REQUIRE(im->depth == MDISP_8U && (im->channels == 1 || im->channels == 3), MDISP_E_UNSUPPORTED,
               "only CV_8UC1 / CV_8UC3 images are supported (got depth %d, %d channels)", im->depth, im->channels);
    return md_h2d_2d(c, d, dpitch, im->data, im->step_bytes, (size_t)im->cols * im->channels, im->rows);
}

extern "C" int mdisp_stereosgbm_stages(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_stereosgbm_params* p,
                                       mdisp_image* out, int16_t* cost, mdisp_image* raw)
{
    MD_REQUIRE(c && p, MDISP_E_BADARG, "null argument");
    int rc;
    if ((rc = check_image(L, "left")) || (rc = check_image(R, "right"))) return rc;
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols && L->channels == R->channels && L->depth == R->depth, MDISP_E_BADARG,
               "left/right size or type mismatch");
    if ((rc = check_out(out, L->rows, L->cols, MDISP_16S, 1, "out"))) return rc;
    if (raw && (rc = check_out(raw, L->rows, L->cols, MDISP_16S, 1, "raw"))) return rc;
    MD_REQUIRE(p->numDisparities > 0 && p->minDisparity >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols, cn = L->channels;
    const int W1 = W - p->minDisparity - p->numDisparities;
    const size_t pitch = md_align((size_t)W * cn, 256), V = W1 > 0 ? (size_t)H * W1 * p->numDisparities : 0;
    Carver cv;
    size_t o_ws = cv.take(md_stereosgbm_bytes(H, W, cn, p)), o_l = cv.take(pitch * H), o_r = cv.take(pitch * H),
           o_d = cv.take((size_t)H * W * 2), o_raw = cv.take((size_t)H * W * 2), o_c = cv.take(cost ? V * 2 : 0);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    if ((rc = upload_u8(c, L, base + o_l, pitch)) || (rc = upload_u8(c, R, base + o_r, pitch))) return rc;
    if ((rc = md_stereosgbm_run(c->stream, base + o_l, base + o_r, pitch, H, W, cn, p, (int16_t*)(base + o_d), base + o_ws,
                                cost ? (int16_t*)(base + o_c) : nullptr, raw ? (int16_t*)(base + o_raw) : nullptr)))
        return rc;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 2, (size_t)W * 2, H));
    if (raw) MD_TRY(md_d2h_2d(c, raw->data, raw->step_bytes, base + o_raw, (size_t)W * 2, (size_t)W * 2, H));
    if (cost && V) MD_HIP(hipMemcpyAsync(cost, base + o_c, V * 2, hipMemcpyDeviceToHost, c->stream));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_stereosgbm_compute(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_stereosgbm_params* p,
                                        mdisp_image* out)
{
    return mdisp_stereosgbm_stages(c, L, R, p, out, nullptr, nullptr);
}

extern "C" size_t mdisp_stereosgbm_workspace_bytes(int rows, int cols, int channels, const mdisp_stereosgbm_params* p)
{
    if (!p || rows <= 0 || cols <= 0 || p->numDisparities <= 0 || p->minDisparity < 0) return 0;
    return md_stereosgbm_bytes(rows, cols, channels, p);
}

extern "C" int mdisp_stereosgbm_dev(mdisp_ctx* c, const uint8_t* dL, const uint8_t* dR, int rows, int cols, int channels, size_t pitch,
                                    const mdisp_stereosgbm_params* p, int16_t* dDisp, void* ws, size_t ws_bytes)
{
    MD_REQUIRE(c && dL && dR && dDisp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(rows > 0 && cols > 0 && (channels == 1 || channels == 3) && pitch >= (size_t)cols * channels, MDISP_E_BADARG, "bad geometry");
    MD_REQUIRE(p->numDisparities > 0 && p->minDisparity >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_HIP(hipSetDevice(c->device));
    const size_t need = md_stereosgbm_bytes(rows, cols, channels, p);
    if (ws) MD_REQUIRE(ws_bytes >= need, MDISP_E_NOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
    else if (!(ws = mdisp_ws(c, need))) return MDISP_E_NOMEM;
    return md_stereosgbm_run(c->stream, dL, dR, pitch, rows, cols, channels, p, dDisp, ws, nullptr, nullptr);
}

// Pairs of a batch side by side: a single pair's launches are partly bound by the serial chains of their scan lines
// (a horizontal direction has one line per image row: about one wave per SIMD) and partly by memory, so up to
// MD_SGBM_LANES pairs run at once, each on a stream and in a workspace slice of its own, and fill each other's gaps.
#define MD_SGBM_LANES 4
static int sgbm_lanes(int n)
{
    const int t = md_tune(MDISP_TUNE_SGBM_BATCH_LANES);
    const int lanes = t > 0 ? (t < MD_SGBM_LANES ? t : MD_SGBM_LANES) : MD_SGBM_LANES;
    return n < lanes ? n : lanes;
}
extern "C" size_t mdisp_stereosgbm_batch_workspace_bytes(int rows, int cols, int channels, const mdisp_stereosgbm_params* p, int n_pairs)
{
    if (!p || rows <= 0 || cols <= 0 || n_pairs <= 0 || p->numDisparities <= 0 || p->minDisparity < 0) return 0;
    const int lanes = n_pairs < MD_SGBM_LANES ? n_pairs : MD_SGBM_LANES;
    return md_align(md_stereosgbm_bytes(rows, cols, channels, p), 256) * lanes;
}

extern "C" int mdisp_stereosgbm_batch_dev(mdisp_ctx* c, const uint8_t* dL, const uint8_t* dR, int n_pairs, size_t pair_stride_bytes,
                                          int rows, int cols, int channels, size_t pitch, const mdisp_stereosgbm_params* p,
                                          int16_t* dDisp, void* ws, size_t ws_bytes)
{
    MD_REQUIRE(c && dL && dR && dDisp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(n_pairs > 0 && rows > 0 && cols > 0 && (channels == 1 || channels == 3) && pitch >= (size_t)cols * channels &&
               pair_stride_bytes >= pitch * rows, MDISP_E_BADARG, "bad geometry");
    MD_REQUIRE(p->numDisparities > 0 && p->minDisparity >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_HIP(hipSetDevice(c->device));
    const size_t one = md_align(md_stereosgbm_bytes(rows, cols, channels, p), 256);
    const size_t need = mdisp_stereosgbm_batch_workspace_bytes(rows, cols, channels, p, n_pairs);
    if (ws) MD_REQUIRE(ws_bytes >= need, MDISP_E_NOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
    else if (!(ws = mdisp_ws(c, need))) return MDISP_E_NOMEM;
    const int lanes = sgbm_lanes(n_pairs);
    if (lanes > 1 && !c->lanes_ready) {
        for (int i = 0; i < 3; i++) MD_HIP(hipStreamCreateWithFlags(&c->lane[i], hipStreamNonBlocking));
        for (int i = 0; i < 4; i++) MD_HIP(hipEventCreateWithFlags(&c->lane_ev[i], hipEventDisableTiming));
        c->lanes_ready = true;
    }
    // the other lanes start after everything enqueued on the context's stream so far, and the stream continues after them
    if (lanes > 1) {
        MD_HIP(hipEventRecord(c->lane_ev[0], c->stream));
        for (int l = 1; l < lanes; l++) MD_HIP(hipStreamWaitEvent(c->lane[l - 1], c->lane_ev[0], 0));
    }
    int rc = MDISP_OK;
    for (int i = 0; i < n_pairs && rc == MDISP_OK; i++) {
        const int l = i % lanes;
        rc = md_stereosgbm_run(l == 0 ? c->stream : c->lane[l - 1], dL + (size_t)i * pair_stride_bytes, dR + (size_t)i * pair_stride_bytes, pitch,
                               rows, cols, channels, p, dDisp + (size_t)i * rows * cols, (u8*)ws + (size_t)l * one, nullptr, nullptr);
    }
    // joined even after a failure, so that the context's stream never runs ahead of work already enqueued on a lane
    for (int l = 1; l < lanes; l++) {
        MD_HIP(hipEventRecord(c->lane_ev[l], c->lane[l - 1]));
        MD_HIP(hipStreamWaitEvent(c->stream, c->lane_ev[l], 0));
    }
    return rc;
}

extern "C" size_t mdisp_stereobm_workspace_bytes(int rows, int cols, const mdisp_stereobm_params* p)
{
    if (!p || rows <= 0 || cols <= 0 || md_stereobm_check(rows, cols, p)) return 0;
    return md_stereobm_bytes(rows, cols, p);
}

extern "C" int mdisp_stereobm_dev(mdisp_ctx* c, const uint8_t* dL, const uint8_t* dR, int rows, int cols, size_t pitch,
                                  const mdisp_stereobm_params* p, int16_t* dDisp, void* ws, size_t ws_bytes)
{
    MD_REQUIRE(c && dL && dR && dDisp && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(rows > 0 && cols > 0 && pitch >= (size_t)cols, MDISP_E_BADARG, "bad geometry");
    MD_TRY(md_stereobm_check(rows, cols, p));
    MD_HIP(hipSetDevice(c->device));
    const size_t need = md_stereobm_bytes(rows, cols, p);
    if (ws) MD_REQUIRE(ws_bytes >= need, MDISP_E_NOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
    else if (!(ws = mdisp_ws(c, need))) return MDISP_E_NOMEM;
    return md_stereobm_run(c->stream, dL, dR, pitch, rows, cols, p, dDisp, ws);
}

extern "C" int mdisp_stereobm_compute(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_stereobm_params* p,
                                      mdisp_image* out)
{
    MD_REQUIRE(c && p, MDISP_E_BADARG, "null argument");
    int rc;
    if ((rc = check_image(L, "left")) || (rc = check_image(R, "right"))) return rc;
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG, "left/right size mismatch");
    // the matcher itself takes CV_8UC1; getDisparity_BM converts colour input with COLOR_BGR2GRAY first (methods.cpp:1850-1860)
    if ((rc = check_out(out, L->rows, L->cols, MDISP_16S, 1, "out"))) return rc;
    const int H = L->rows, W = L->cols;
    MD_TRY(md_stereobm_check(H, W, p));
    MD_HIP(hipSetDevice(c->device));
    const size_t pitch = md_align((size_t)W, 256), p3 = md_align((size_t)W * 3, 256);
    const bool colour = L->channels == 3 || R->channels == 3;
    Carver cv;
    size_t o_ws = cv.take(md_stereobm_bytes(H, W, p)), o_l = cv.take(pitch * H), o_r = cv.take(pitch * H), o_d = cv.take((size_t)H * W * 2),
           o_st = cv.take(colour ? p3 * H : 0);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    if ((rc = upload_gray(c, L, base + o_l, pitch, base + o_st, p3)) || (rc = upload_gray(c, R, base + o_r, pitch, base + o_st, p3))) return rc;
    MD_TRY(md_stereobm_run(c->stream, base + o_l, base + o_r, pitch, H, W, p, (int16_t*)(base + o_d), base + o_ws));
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 2, (size_t)W * 2, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_fgs_filter(mdisp_ctx* c, const mdisp_image* guide, const mdisp_image* src, double lambda, double sigma_color,
                                double lambda_attenuation, int num_iter, mdisp_image* dst)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(guide, "guide")) || (rc = check_image(src, "src"))) return rc;
    MD_REQUIRE(guide->depth == MDISP_8U && (guide->channels == 1 || guide->channels == 3), MDISP_E_BADARG, "guide must be CV_8UC1 or CV_8UC3");
    MD_REQUIRE((src->depth == MDISP_32F || src->depth == MDISP_16S) && src->channels == 1, MDISP_E_UNSUPPORTED,
               "the filter is built for CV_32FC1 and CV_16SC1 maps");
    MD_REQUIRE(src->rows == guide->rows && src->cols == guide->cols, MDISP_E_BADARG, "guide / src size mismatch");
    if ((rc = check_out(dst, src->rows, src->cols, src->depth, 1, "dst"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = src->rows, W = src->cols, cn = guide->channels;
    const size_t gp = md_align((size_t)W * cn + cn, 256);
    Carver cv;
    size_t o_ws = cv.take(md_fgs_bytes(H, W)), o_g = cv.take(gp * (H + 1)), o_u = cv.take((size_t)H * W * 4), o_s = cv.take((size_t)H * W * 2);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_g, gp, guide->data, guide->step_bytes, (size_t)W * cn, H));
    float* u = (float*)(base + o_u);
    if (src->depth == MDISP_32F) MD_TRY(md_h2d_2d(c, u, (size_t)W * 4, src->data, src->step_bytes, (size_t)W * 4, H));
    else {
        MD_TRY(md_h2d_2d(c, base + o_s, (size_t)W * 2, src->data, src->step_bytes, (size_t)W * 2, H));
        MD_TRY(md_fgs_convert(c->stream, (const int16_t*)(base + o_s), u, (size_t)H * W));
    }
    MD_TRY(md_fgs_filter(c->stream, base + o_g, gp, H, W, cn, u, lambda, sigma_color, lambda_attenuation, num_iter, base + o_ws));
    if (src->depth == MDISP_32F) MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, u, (size_t)W * 4, (size_t)W * 4, H));
    else {
        MD_TRY(md_fgs_convert_back(c->stream, u, (int16_t*)(base + o_s), (size_t)H * W));
        MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, base + o_s, (size_t)W * 2, (size_t)W * 2, H));
    }
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_median_blur3_s16(mdisp_ctx* c, const mdisp_image* src, mdisp_image* dst)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    MD_REQUIRE(src->depth == MDISP_16S && src->channels == 1, MDISP_E_UNSUPPORTED, "medianBlur(3) is built for CV_16SC1");
    if ((rc = check_out(dst, src->rows, src->cols, MDISP_16S, 1, "dst"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = src->rows, W = src->cols;
    const size_t n = md_align((size_t)H * W * 2, 256);
    u8* base = (u8*)mdisp_ws(c, 2 * n);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base, (size_t)W * 2, src->data, src->step_bytes, (size_t)W * 2, H));
    MD_TRY(md_median3_s16(c->stream, (const int16_t*)base, H, W, (int16_t*)(base + n)));
    MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, base + n, (size_t)W * 2, (size_t)W * 2, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_filter_speckles_s16(mdisp_ctx* c, mdisp_image* img, int newVal, int maxSpeckleSize, int maxDiff)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(img, "img"))) return rc;
    MD_REQUIRE(img->depth == MDISP_16S && img->channels == 1, MDISP_E_UNSUPPORTED, "filterSpeckles is built for CV_16SC1");
    MD_REQUIRE((size_t)img->rows * img->cols < ((size_t)1 << 30), MDISP_E_RANGE, "image too large");
    MD_HIP(hipSetDevice(c->device));
    const int H = img->rows, W = img->cols;
    const size_t n = md_align((size_t)H * W * 2, 256);
    u8* base = (u8*)mdisp_ws(c, n + md_speckle_scratch_bytes(H, W));
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base, (size_t)W * 2, img->data, img->step_bytes, (size_t)W * 2, H));
    MD_TRY(md_filter_speckles_s16(c->stream, (int16_t*)base, H, W, newVal, maxSpeckleSize, maxDiff, base + n));
    MD_TRY(md_d2h_2d(c, img->data, img->step_bytes, base, (size_t)W * 2, (size_t)W * 2, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_sgbm_volume(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, const mdisp_sgbm_params* p, uint16_t* Sh)
{
    MD_REQUIRE(c && Sh, MDISP_E_BADARG, "null argument");
    int rc;
    if ((rc = check_image(L, "left")) || (rc = check_image(R, "right"))) return rc;
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG, "left/right size mismatch");
    md_sgm_plan pl;
    if ((rc = md_sgm_plan_make(L->rows, L->cols, p, 1, &pl, true))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = L->rows, W = L->cols;
    const size_t pitch = md_align(W, 256), p3 = md_align((size_t)W * 3, 256);
    Carver cv;
    size_t o_ws = cv.take(pl.total), o_l = cv.take(pitch * H), o_r = cv.take(pitch * H), o_st = cv.take(p3 * H);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    if ((rc = upload_gray(c, L, base + o_l, pitch, base + o_st, p3))) return rc;
    if ((rc = upload_gray(c, R, base + o_r, pitch, base + o_st, p3))) return rc;
    const u16* S;
    if ((rc = md_sgm_aggregate_only(c, pl, base + o_l, base + o_r, pitch, base + o_ws, &S))) return rc;
    MD_TRY(md_d2h_2d(c, Sh, (size_t)pl.D * 2, S, (size_t)pl.sstride * 2, (size_t)pl.D * 2, (size_t)H * W));
    MD_HIP(hipStreamSynchronize(c->stream));
    return md_check_sgm_error(c);
}

// ---------------------------------------------------------------------------------------------
// LR check, remap, rectification maps

extern "C" int mdisp_lr_check(mdisp_ctx* c, const mdisp_image* dL, const mdisp_image* dR, float thresh,
                              mdisp_image* validL, mdisp_image* validR)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_image(dL, "dispL")) || (rc = check_image(dR, "dispR"))) return rc;
    MD_REQUIRE(dL->depth == MDISP_32F && dR->depth == MDISP_32F && dL->channels == 1 && dR->channels == 1 &&
               dL->rows == dR->rows && dL->cols == dR->cols, MDISP_E_BADARG, "LR check needs two CV_32FC1 maps of equal size");
    if ((rc = check_out(validL, dL->rows, dL->cols, MDISP_8U, 1, "validL")) || (rc = check_out(validR, dL->rows, dL->cols, MDISP_8U, 1, "validR"))) return rc;
    MD_HIP(hipSetDevice(c->device));
    const int H = dL->rows, W = dL->cols;
    Carver cv;
    size_t o_l = cv.take((size_t)H * W * 4), o_r = cv.take((size_t)H * W * 4), o_vl = cv.take((size_t)H * W), o_vr = cv.take((size_t)H * W);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    hipStream_t s = c->stream;
    MD_TRY(md_h2d_2d(c, base + o_l, (size_t)W * 4, dL->data, dL->step_bytes, (size_t)W * 4, H));
    MD_TRY(md_h2d_2d(c, base + o_r, (size_t)W * 4, dR->data, dR->step_bytes, (size_t)W * 4, H));
    if ((rc = md_lr_check(s, (float*)(base + o_l), (float*)(base + o_r), H, W, thresh, base + o_vl, base + o_vr))) return rc;
    MD_TRY(md_d2h_2d(c, validL->data, validL->step_bytes, base + o_vl, W, W, H));
    MD_TRY(md_d2h_2d(c, validR->data, validR->step_bytes, base + o_vr, W, W, H));
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}

// ---- device-resident rectification maps: built once per calibration on the GPU, reused for every frame ----
struct mdisp_rectify_maps { int device, W, H; int16_t* m1; u16* m2; };

extern "C" int mdisp_rectify_maps_create(mdisp_ctx* c, const double K[9], const double* dist, int ndist, const double R[9],
                                         const double P[12], int width, int height, mdisp_rectify_maps** out)
{
    MD_REQUIRE(c && out && width > 0 && height > 0, MDISP_E_BADARG, "null argument or empty size");
    *out = nullptr;
    double ir[9], k[8];
    MD_TRY(md_rectify_map_setup(K, dist, ndist, R, P, ir, k));
    MD_HIP(hipSetDevice(c->device));
    mdisp_rectify_maps* m = new (std::nothrow) mdisp_rectify_maps();
    MD_REQUIRE(m, MDISP_E_NOMEM, "out of host memory");
    m->device = c->device; m->W = width; m->H = height; m->m1 = nullptr; m->m2 = nullptr;
    const size_t n = (size_t)width * height;
    if (hipMalloc((void**)&m->m1, n * 4) != hipSuccess || hipMalloc((void**)&m->m2, n * 2) != hipSuccess) {
        if (m->m1) (void)hipFree(m->m1);
        delete m;
        mdisp_set_error("hipMalloc of the rectification maps (%zu bytes) failed", n * 6);
        return MDISP_E_NOMEM;
    }
    int rc = md_rectify_maps(c->stream, ir, k, K[0], K[4], K[2], K[5], width, height, m->m1, m->m2);
    if (rc == MDISP_OK && hipStreamSynchronize(c->stream) != hipSuccess) { mdisp_set_error("rectification map kernel failed"); rc = MDISP_E_HIP; }
    if (rc != MDISP_OK) { (void)hipFree(m->m1); (void)hipFree(m->m2); delete m; return rc; }
    *out = m;
    return MDISP_OK;
}

extern "C" void mdisp_rectify_maps_destroy(mdisp_rectify_maps* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(m->m1); (void)hipFree(m->m2);
    delete m;
}

extern "C" int mdisp_rectify_maps_download(mdisp_ctx* c, const mdisp_rectify_maps* m, int16_t* map1, uint16_t* map2)
{
    MD_REQUIRE(c && m && map1 && map2, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(m->device == c->device, MDISP_E_BADARG, "maps live on device %d, context on %d", m->device, c->device);
    MD_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)m->W * m->H;
    MD_HIP(hipMemcpyAsync(map1, m->m1, n * 4, hipMemcpyDeviceToHost, c->stream));
    MD_HIP(hipMemcpyAsync(map2, m->m2, n * 2, hipMemcpyDeviceToHost, c->stream));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_remap_dev(mdisp_ctx* c, const mdisp_rectify_maps* m, const uint8_t* d_src, size_t src_pitch, int src_rows, int src_cols,
                               int channels, uint8_t* d_dst, size_t dst_pitch)
{
    MD_REQUIRE(c && m && d_src && d_dst, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(m->device == c->device, MDISP_E_BADARG, "maps live on device %d, context on %d", m->device, c->device);
    MD_REQUIRE(src_rows > 0 && src_cols > 0 && (channels == 1 || channels == 3) && src_pitch >= (size_t)src_cols * channels &&
               dst_pitch >= (size_t)m->W * channels, MDISP_E_BADARG, "bad geometry");
    MD_HIP(hipSetDevice(c->device));
    return md_remap(c->stream, d_src, src_pitch, src_rows, src_cols, channels, m->m1, m->m2, m->H, m->W, d_dst, dst_pitch);
}

extern "C" int mdisp_remap_with(mdisp_ctx* c, const mdisp_rectify_maps* m, const mdisp_image* src, mdisp_image* dst)
{
    MD_REQUIRE(c && m, MDISP_E_BADARG, "null argument");
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    MD_REQUIRE(src->depth == MDISP_8U && dst->depth == MDISP_8U && src->channels == dst->channels &&
               (src->channels == 1 || src->channels == 3), MDISP_E_BADARG, "remap needs CV_8UC1 or CV_8UC3 images");
    MD_REQUIRE(dst->rows == m->H && dst->cols == m->W, MDISP_E_BADARG, "dst must have the size of the maps (%dx%d)", m->W, m->H);
    MD_REQUIRE(m->device == c->device, MDISP_E_BADARG, "maps live on device %d, context on %d", m->device, c->device);
    MD_HIP(hipSetDevice(c->device));
    const int sH = src->rows, sW = src->cols, cn = src->channels;
    const size_t sp = md_align((size_t)sW * cn, 256), dp = md_align((size_t)m->W * cn, 256);
    Carver cv;
    size_t o_s = cv.take(sp * sH), o_d = cv.take(dp * m->H);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_s, sp, src->data, src->step_bytes, (size_t)sW * cn, sH));
    if ((rc = md_remap(c->stream, base + o_s, sp, sH, sW, cn, m->m1, m->m2, m->H, m->W, base + o_d, dp))) return rc;
    MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, base + o_d, dp, (size_t)m->W * cn, m->H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_remap(mdisp_ctx* c, const mdisp_image* src, const int16_t* map1, const uint16_t* map2, mdisp_image* dst)
{
    MD_REQUIRE(c && map1 && map2, MDISP_E_BADARG, "null argument");
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    MD_REQUIRE(src->depth == MDISP_8U && dst->depth == MDISP_8U && src->channels == dst->channels &&
               (src->channels == 1 || src->channels == 3), MDISP_E_BADARG, "remap needs CV_8UC1 or CV_8UC3 images");
    MD_HIP(hipSetDevice(c->device));
    const int sH = src->rows, sW = src->cols, dH = dst->rows, dW = dst->cols, cn = src->channels;
    const size_t sp = md_align((size_t)sW * cn, 256), dp = md_align((size_t)dW * cn, 256);
    Carver cv;
    size_t o_s = cv.take(sp * sH), o_d = cv.take(dp * dH), o_m1 = cv.take((size_t)dH * dW * 4), o_m2 = cv.take((size_t)dH * dW * 2);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    hipStream_t s = c->stream;
    MD_TRY(md_h2d_2d(c, base + o_s, sp, src->data, src->step_bytes, (size_t)sW * cn, sH));
    MD_HIP(hipMemcpyAsync(base + o_m1, map1, (size_t)dH * dW * 4, hipMemcpyHostToDevice, s));
    MD_HIP(hipMemcpyAsync(base + o_m2, map2, (size_t)dH * dW * 2, hipMemcpyHostToDevice, s));
    if ((rc = md_remap(s, base + o_s, sp, sH, sW, cn, (const int16_t*)(base + o_m1), (const u16*)(base + o_m2), dH, dW, base + o_d, dp))) return rc;
    MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, base + o_d, dp, (size_t)dW * cn, dH));
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}
