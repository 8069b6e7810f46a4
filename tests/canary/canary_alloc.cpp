// Guard-zone device allocator for the GPU test suite (TEST INFRASTRUCTURE — never loaded by the product).
//
// The kernels of libvidi_hip.so replace bounds predicates by buffer-descriptor range checks, hand-counted vmcnt waits and LDS rings
// (gemm_w4.h, attn_cross_rows.hip, gemv_mfma.hip): a wrong descriptor range or row count writes past a tensor without any functional
// symptom as long as the caching allocator happens to own the neighbouring bytes.  With VIDI_CANARY=1 tests/conftest.py installs this
// file as torch's pluggable device allocator: EVERY tensor (test inputs and outputs, the engine's workspaces, K/V caches) becomes its own
// hipMalloc with a poisoned zone in front (4 KiB) and behind (256 KiB, + the padding up to the next 256 bytes).  The zones are compared
// with the pattern when the tensor is freed and — through canary_check_all(), called by a pytest fixture after every test, or after every
// C-ABI call with VIDI_CANARY=2 — while it is alive.  A stray store of up to 256 KiB past either end of any tensor fails the test that made it.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
constexpr size_t PRE = 4096, POST = 256 * 1024;
constexpr unsigned char PATTERN = 0xA5;
struct Rec { char* base; size_t size, padded; int device; unsigned long serial; };
std::mutex mu;
std::unordered_map<void*, Rec> live;
std::atomic<long> violations{0};
std::atomic<unsigned long> serial{0};
char first_msg[512] = {0};
std::vector<unsigned char> host;

// compares both zones of one allocation with the pattern (caller holds the lock, the device is idle); returns the number of damaged bytes
long check_rec(const Rec& r) {
    size_t tail = r.padded - r.size + POST;
    if (host.size() < PRE + tail) host.resize(PRE + tail);
    if (hipMemcpy(host.data(), r.base, PRE, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (hipMemcpy(host.data() + PRE, r.base + PRE + r.size, tail, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    long bad = 0, first = -1;
    for (size_t i = 0; i < PRE + tail; ++i)
        if (host[i] != PATTERN) { if (first < 0) first = (long)i; ++bad; }
    if (bad) {
        long before = violations.fetch_add(1);
        if (before == 0)
            snprintf(first_msg, sizeof first_msg, "allocation #%lu of %zu bytes: %ld guard byte(s) overwritten, first at offset %ld %s the tensor",
                     r.serial, r.size, bad, first < (long)PRE ? (long)PRE - first : first - (long)PRE, first < (long)PRE ? "BEFORE the start of" : "PAST the end of");
        // re-arm, so that one stray store is reported once
        hipMemset(r.base, PATTERN, PRE);
        hipMemset(r.base + PRE + r.size, PATTERN, tail);
        hipDeviceSynchronize();
    }
    return bad;
}
}  // namespace

extern "C" {

void* canary_malloc(ssize_t size, int device, hipStream_t) {
    int prev = -1;
    hipGetDevice(&prev);
    if (prev != device) hipSetDevice(device);
    size_t sz = size > 0 ? (size_t)size : 0, padded = (sz + 255) & ~(size_t)255;
    char* base = nullptr;
    if (hipMalloc((void**)&base, PRE + padded + POST) != hipSuccess) {
        if (prev != device && prev >= 0) hipSetDevice(prev);
        return nullptr;
    }
    hipMemset(base, PATTERN, PRE);
    hipMemset(base + PRE + sz, PATTERN, padded - sz + POST);
    hipDeviceSynchronize();          // torch's streams do not wait for the null stream: the zones must be in place before anything runs
    {
        std::lock_guard<std::mutex> g(mu);
        live[base + PRE] = Rec{base, sz, padded, device, ++serial};
    }
    if (prev != device && prev >= 0) hipSetDevice(prev);
    return base + PRE;
}

void canary_free(void* ptr, ssize_t, int device, hipStream_t) {
    if (!ptr) return;
    int prev = -1;
    hipGetDevice(&prev);
    if (prev != device) hipSetDevice(device);
    hipDeviceSynchronize();
    Rec r{};
    bool found = false;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = live.find(ptr);
        if (it != live.end()) { r = it->second; live.erase(it); found = true; check_rec(r); }
    }
    if (found) hipFree(r.base);
    if (prev != device && prev >= 0) hipSetDevice(prev);
}

// checks the zones of every live allocation; -> violations seen since the process started (msg: the first one)
long canary_check_all(char* msg, int n) {
    hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> g(mu);
        int prev = -1;
        hipGetDevice(&prev);
        for (auto& kv : live) {
            if (kv.second.device != prev) hipSetDevice(kv.second.device);
            check_rec(kv.second);
            if (kv.second.device != prev && prev >= 0) hipSetDevice(prev);
        }
    }
    if (msg && n > 0) { strncpy(msg, first_msg, (size_t)n - 1); msg[n - 1] = 0; }
    return violations.load();
}

long canary_live_allocations() { std::lock_guard<std::mutex> g(mu); return (long)live.size(); }
long canary_total_allocations() { return (long)serial.load(); }
void canary_reset() { violations.store(0); first_msg[0] = 0; }

}  // extern "C"
