// Host-side multi-GPU entry points of the C ABI: one host thread and one context per device, pinned staging, no
// collective.  The reference is a single-process program whose only concurrency is two std::async calls
// (cameraCalibration/cameraCalibration.cpp:163-190); these entry points are what a caller of that shape uses to spread a
// batch of pairs -- or the rows of one large pair -- over the GPUs of a node.
//
//   mdisp_sgbm_batch_multi   pairs are independent: every device gets a contiguous block of the batch and runs the chunked,
//                            double-buffered mdisp_sgbm_batch on it (uploads of chunk k+1 and downloads of chunk k-1 behind
//                            the kernels of chunk k).  With MDISP_MULTI_PIN_HOST the caller's images and maps are page-locked
//                            for the duration of the call, so those copies are real DMA transfers that overlap the kernels.
//   mdisp_sgbm_rows_multi    one pair, row ranges per device (mdisp_sgbm_rows_sweep_dev): both views go to every device, the
//                            downward wave-front starts at the first device and the upward one at the last at the same time,
//                            and the path states of a boundary row travel device to device with hipMemcpyPeerAsync (a plain
//                            device copy when both participants share a GPU).
#include "common.h"
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct WorkerResult { int rc = MDISP_OK; std::string msg; };

void block_of(int total, int rank, int world, int* first, int* count)
{
    const int base = total / world, extra = total % world;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + std::min(rank, extra);
}

// page-locks a set of host ranges for the lifetime of the object.  Registration works on whole pages and refuses a page
// twice, so ranges are rounded to pages and merged where they touch or overlap (the images of a batch often lie back to
// back in one allocation) before they are registered; ranges that cannot be registered stay pageable.
struct PinnedRanges {
    std::vector<std::pair<uintptr_t, uintptr_t>> want;
    std::vector<void*> held;
    void add(const void* p, size_t bytes)
    {
        if (!p || bytes == 0) return;
        const uintptr_t page = 4096, a = (uintptr_t)p & ~(page - 1), b = ((uintptr_t)p + bytes + page - 1) & ~(page - 1);
        want.emplace_back(a, b);
    }
    void commit()
    {
        std::sort(want.begin(), want.end());
        size_t i = 0;
        while (i < want.size()) {
            uintptr_t a = want[i].first, b = want[i].second;
            size_t k = i + 1;
            while (k < want.size() && want[k].first <= b) { b = std::max(b, want[k].second); k++; }
            if (hipHostRegister((void*)a, b - a, hipHostRegisterDefault) == hipSuccess) held.push_back((void*)a);
            else (void)hipGetLastError();
            i = k;
        }
        want.clear();
    }
    ~PinnedRanges() { for (void* p : held) (void)hipHostUnregister(p); }
};

// The workers' contexts live as long as the process: a context owns a grow-only workspace (gigabytes for a chunk of
// full-size pairs), and allocating it anew in every call costs more than the copies the call overlaps.  Worker r of a call
// uses the r-th context created for its device; one multi-device call runs at a time.
std::mutex g_multi_mu;
std::vector<std::pair<int, mdisp_ctx*>> g_multi_ctx;      // (device, context), in creation order
mdisp_ctx* worker_ctx(int device, int nth, int* rc)      // the nth context of `device` (created on demand); g_multi_mu held by the caller's call
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int seen = 0;
    for (auto& e : g_multi_ctx)
        if (e.first == device && seen++ == nth) return e.second;
    mdisp_ctx* c = nullptr;
    *rc = mdisp_create(&c, device, nullptr);
    if (*rc == MDISP_OK) g_multi_ctx.emplace_back(device, c);
    return c;
}
int nth_of(const int* devices, int r)
{
    int n = 0;
    for (int i = 0; i < r; i++) n += devices[i] == devices[r];
    return n;
}

int finish(const std::vector<WorkerResult>& res, const char* what)
{
    for (size_t i = 0; i < res.size(); i++)
        if (res[i].rc != MDISP_OK) {
            mdisp_set_error("%s: worker %zu: %s", what, i, res[i].msg.c_str());
            return res[i].rc;
        }
    return MDISP_OK;
}

}  // namespace

extern "C" int mdisp_sgbm_batch_multi(const int* devices, int ndevices, const uint8_t* const* lefts, const uint8_t* const* rights, int npairs,
                                      int rows, int cols, size_t step_bytes, const mdisp_sgbm_params* p, int16_t* const* disps, int chunk,
                                      int flags)
{
    MD_REQUIRE(devices && ndevices > 0 && lefts && rights && disps && p, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(npairs >= 0 && rows > 0 && cols > 0 && step_bytes >= (size_t)cols, MDISP_E_BADARG, "bad geometry");
    const int present = mdisp_device_count();
    if (present <= 0) { mdisp_set_error("no HIP device available; libmdisp has no CPU fallback"); return MDISP_E_NODEVICE; }
    for (int i = 0; i < ndevices; i++) MD_REQUIRE(devices[i] >= 0 && devices[i] < present, MDISP_E_BADARG, "device %d out of range (%d present)", devices[i], present);
    if (npairs == 0) return MDISP_OK;
    for (int i = 0; i < npairs; i++) MD_REQUIRE(lefts[i] && rights[i] && disps[i], MDISP_E_BADARG, "pair %d: null image", i);
    const int world = std::min(ndevices, npairs);
    std::lock_guard<std::mutex> oneCall(g_multi_mu);
    std::vector<WorkerResult> res(world);
    std::vector<std::thread> th;
    const size_t imgBytes = (size_t)(rows - 1) * step_bytes + cols, outBytes = (size_t)rows * cols * 2;
    auto work = [&](int r) {
            int first, count;
            block_of(npairs, r, world, &first, &count);
            WorkerResult& out = res[r];
            if (hipSetDevice(devices[r]) != hipSuccess) { out.rc = MDISP_E_NODEVICE; out.msg = "hipSetDevice failed"; return; }
            PinnedRanges pins;
            if (flags & MDISP_MULTI_PIN_HOST)
            {
                for (int i = first; i < first + count; i++) { pins.add(lefts[i], imgBytes); pins.add(rights[i], imgBytes); pins.add(disps[i], outBytes); }
                pins.commit();
            }
            mdisp_ctx* ctx = worker_ctx(devices[r], nth_of(devices, r), &out.rc);
            if (out.rc == MDISP_OK) {
                out.rc = mdisp_sgbm_batch(ctx, lefts + first, rights + first, count, rows, cols, step_bytes, p, disps + first, chunk);
                if (out.rc == MDISP_OK) out.rc = mdisp_synchronize(ctx);
            }
            if (out.rc != MDISP_OK) out.msg = mdisp_last_error();      // the message is per thread: carry it to the caller
    };
    // the last worker runs on the calling thread (a single worker costs no thread at all); the caller's current device is restored
    int callerDevice = 0;
    const bool haveCaller = hipGetDevice(&callerDevice) == hipSuccess;
    for (int r = 0; r + 1 < world; r++) th.emplace_back(work, r);
    work(world - 1);
    for (auto& t : th) t.join();
    if (haveCaller) (void)hipSetDevice(callerDevice);
    return finish(res, "mdisp_sgbm_batch_multi");
}

namespace {

// what the participants of a row-sharded pair share: each one's buffers, and which boundary states have arrived
struct RowsShared {
    std::mutex mu;
    std::condition_variable cv;
    int allocated = 0;
    bool failed = false;
    std::vector<u8*> bndIn[2];           // [direction][rank]: where the neighbour delivers the states this rank starts from
    std::vector<char> arrived[2];        // [direction][rank]
};

}  // namespace

extern "C" int mdisp_sgbm_rows_multi(const int* devices, int ndevices, const mdisp_image* L, const mdisp_image* R,
                                     const mdisp_sgbm_params* p, mdisp_image* out)
{
    MD_REQUIRE(devices && ndevices > 0 && L && R && p && out, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(L->data && R->data && out->data && L->rows > 0 && L->cols > 0 && L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG,
               "left/right size mismatch or empty image");
    MD_REQUIRE(L->depth == MDISP_8U && R->depth == MDISP_8U && L->channels == 1 && R->channels == 1, MDISP_E_UNSUPPORTED,
               "mdisp_sgbm_rows_multi takes CV_8UC1 views");
    MD_REQUIRE(out->rows == L->rows && out->cols == L->cols && out->depth == MDISP_16S && out->channels == 1 &&
               out->step_bytes >= (size_t)L->cols * 2, MDISP_E_BADARG, "out: expected a %dx%d CV_16SC1 image", L->cols, L->rows);
    const int present = mdisp_device_count();
    if (present <= 0) { mdisp_set_error("no HIP device available; libmdisp has no CPU fallback"); return MDISP_E_NODEVICE; }
    for (int i = 0; i < ndevices; i++) MD_REQUIRE(devices[i] >= 0 && devices[i] < present, MDISP_E_BADARG, "device %d out of range (%d present)", devices[i], present);
    const int H = L->rows, W = L->cols, world = ndevices;
    int bandRows = 0;
    size_t bndBytes = 0, wsBytes = 0;
    MD_TRY(mdisp_sgbm_rows_geometry(H, W, p, &bandRows, &bndBytes, &wsBytes));
    const int bands = (H + bandRows - 1) / bandRows;
    MD_REQUIRE(world <= bands, MDISP_E_BADARG, "%d participants for %d row bands of %d rows: some would have no rows", world, bands, bandRows);
    std::lock_guard<std::mutex> oneCall(g_multi_mu);
    RowsShared sh;
    for (int d = 0; d < 2; d++) { sh.bndIn[d].assign(world, nullptr); sh.arrived[d].assign(world, 0); }
    std::vector<WorkerResult> res(world);
    std::vector<std::thread> th;
    const size_t pitch = md_align((size_t)W, 256);
    for (int r = 0; r < world; r++) {
        th.emplace_back([&, r]() {
            WorkerResult& wr = res[r];
            mdisp_ctx* ctx = nullptr;
            u8 *dL = nullptr, *dR = nullptr, *ws = nullptr, *bIn[2] = {nullptr, nullptr}, *bOut = nullptr;
            int16_t* dDisp = nullptr;
            auto fail = [&](int rc, const char* msg) { wr.rc = rc; wr.msg = msg; };
            auto hip = [&](hipError_t e, const char* what) {
                if (e == hipSuccess) return true;
                fail(MDISP_E_HIP, (std::string(what) + " -> " + hipGetErrorString(e)).c_str());
                return false;
            };
            int bFirst, bCount;
            block_of(bands, r, world, &bFirst, &bCount);
            const int rowBegin = std::min(H, bFirst * bandRows), rowEnd = std::min(H, (bFirst + bCount) * bandRows);
            bool ok = hipSetDevice(devices[r]) == hipSuccess;
            if (!ok) fail(MDISP_E_NODEVICE, "hipSetDevice failed");
            if (ok) { ctx = worker_ctx(devices[r], nth_of(devices, r), &wr.rc); if (wr.rc != MDISP_OK) { wr.msg = mdisp_last_error(); ok = false; } }
            ok = ok && hip(hipMalloc((void**)&dL, pitch * H), "hipMalloc") && hip(hipMalloc((void**)&dR, pitch * H), "hipMalloc") &&
                 hip(hipMalloc((void**)&ws, wsBytes), "hipMalloc") && hip(hipMalloc((void**)&bIn[0], bndBytes), "hipMalloc") &&
                 hip(hipMalloc((void**)&bIn[1], bndBytes), "hipMalloc") && hip(hipMalloc((void**)&bOut, bndBytes), "hipMalloc") &&
                 hip(hipMalloc((void**)&dDisp, (size_t)H * W * 2), "hipMalloc");
            if (ok) {
                // (through the context: one linear copy plus a device-side 2-D copy -- a pitched copy from pageable memory runs row by row)
                wr.rc = md_h2d_2d(ctx, dL, pitch, L->data, L->step_bytes, W, H);
                if (wr.rc == MDISP_OK) wr.rc = md_h2d_2d(ctx, dR, pitch, R->data, R->step_bytes, W, H);
                if (wr.rc == MDISP_OK) wr.rc = mdisp_synchronize(ctx);
                if (wr.rc != MDISP_OK) { wr.msg = mdisp_last_error(); ok = false; }
            }
            {   // every participant's receive buffers exist before anybody delivers into them
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.bndIn[0][r] = bIn[0]; sh.bndIn[1][r] = bIn[1];
                if (!ok) sh.failed = true;
                sh.allocated++;
                sh.cv.notify_all();
                sh.cv.wait(lk, [&] { return sh.allocated == world; });
                if (sh.failed && ok) { ok = false; fail(MDISP_E_HIP, "another participant failed to set up"); }
            }
            // the two sweeps in the order their boundary states arrive: the downward wave-front reaches rank r after r hops,
            // the upward one after world - 1 - r
            const int order[2] = {r <= world - 1 - r ? 0 : 1, r <= world - 1 - r ? 1 : 0};
            hipStream_t copy = nullptr;
            ok = ok && hip(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking), "hipStreamCreate");
            for (int step = 0; step < 2; step++) {
                const int dir = order[step];
                const int src = dir == 0 ? r - 1 : r + 1, dst = dir == 0 ? r + 1 : r - 1;
                const bool hasSrc = src >= 0 && src < world, hasDst = dst >= 0 && dst < world;
                if (hasSrc) {
                    std::unique_lock<std::mutex> lk(sh.mu);
                    sh.cv.wait(lk, [&] { return sh.arrived[dir][r] != 0 || sh.failed; });
                    if (sh.failed && ok) { ok = false; fail(MDISP_E_HIP, "another participant failed"); }
                }
                if (ok) {
                    wr.rc = mdisp_sgbm_rows_sweep_dev(ctx, dL, dR, H, W, pitch, p, rowBegin, rowEnd, dir, step, hasSrc ? bIn[dir] : nullptr,
                                                      hasDst ? bOut : nullptr, dDisp, ws, wsBytes);
                    if (wr.rc == MDISP_OK) wr.rc = mdisp_synchronize(ctx);
                    if (wr.rc != MDISP_OK) { wr.msg = mdisp_last_error(); ok = false; }
                }
                if (hasDst) {
                    if (ok) {
                        // device to device over the fabric; the same call is a plain device copy when both ranks share a GPU
                        ok = devices[dst] == devices[r] ? hip(hipMemcpyAsync(sh.bndIn[dir][dst], bOut, bndBytes, hipMemcpyDeviceToDevice, copy), "boundary copy")
                                                        : hip(hipMemcpyPeerAsync(sh.bndIn[dir][dst], devices[dst], bOut, devices[r], bndBytes, copy), "boundary copy");
                        ok = ok && hip(hipStreamSynchronize(copy), "boundary copy");
                    }
                    std::lock_guard<std::mutex> lk(sh.mu);
                    if (!ok) sh.failed = true;
                    sh.arrived[dir][dst] = 1;
                    sh.cv.notify_all();
                }
            }
            if (ok && rowEnd > rowBegin) {
                wr.rc = md_d2h_2d(ctx, (u8*)out->data + (size_t)rowBegin * out->step_bytes, out->step_bytes, dDisp + (size_t)rowBegin * W, (size_t)W * 2,
                                  (size_t)W * 2, rowEnd - rowBegin);
                if (wr.rc == MDISP_OK) wr.rc = mdisp_synchronize(ctx);
                if (wr.rc != MDISP_OK) { wr.msg = mdisp_last_error(); ok = false; }
            }
            if (!ok) { std::lock_guard<std::mutex> lk(sh.mu); sh.failed = true; sh.cv.notify_all(); }
            if (copy) (void)hipStreamDestroy(copy);
            for (void* q : {(void*)dL, (void*)dR, (void*)ws, (void*)bIn[0], (void*)bIn[1], (void*)bOut, (void*)dDisp}) if (q) (void)hipFree(q);
            if (!ok && wr.rc == MDISP_OK) wr.rc = MDISP_E_HIP;
        });
    }
    for (auto& t : th) t.join();
    return finish(res, "mdisp_sgbm_rows_multi");
}
