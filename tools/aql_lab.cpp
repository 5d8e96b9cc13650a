// LAB: kernels dispatched through an HSA queue of our own, AQL packets written by hand, WITHOUT the barrier bit (HIP sets it on every launch of a
// stream and ignores hipExtAnyOrderLaunch on gfx950: profiles/r03_anyorder_lab.log).  Questions: (1) do consecutive packets of one queue overlap,
// (2) are their workgroups still dispatched in packet order (the deadlock-freedom argument of chained launches: a workgroup that waits for the
// previous kernel only ever runs when ALL workgroups of that kernel have been dispatched), (3) what does a dependency through an agent-scope
// release / acquire counter cost from the last producer workgroup's end to the consumers seeing the data, (4) packets per second of a chain.
//   hipcc --offload-arch=gfx950 -O2 --cuda-device-only --no-gpu-bundle-output tools/aql_lab_kernels.hip -o /tmp/aql_lab.hsaco
//   hipcc -O2 tools/aql_lab.cpp -o /tmp/aql_lab -lhsa-runtime64 && /tmp/aql_lab /tmp/aql_lab.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <fstream>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = ""; hsa_status_string(s_, &m_); printf("HSA error 0x%x (%s) at %s:%d\n", s_, m_, __FILE__, __LINE__); exit(1); } } while (0)

static hsa_agent_t g_gpu; static bool g_have = false;
static int g_nofence = 0;   // argv[2] = "nofence": packets without the barrier bit also carry no acquire / release fence
static hsa_status_t pick_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
    return HSA_STATUS_SUCCESS;
}

struct Args {            // lab_kernel's explicit arguments, natural alignment
    unsigned long long* out; int base; int pad0; long long ticks; unsigned* wait; unsigned wait_val; unsigned pad1; unsigned* sig;
    unsigned* data; int ndata; unsigned stamp; unsigned long long* seen; unsigned* bad;
};

struct Queue {
    hsa_queue_t* q; uint64_t kobj; uint32_t group, priv_, kasz;
    void submit(const void* kernarg, uint32_t wgs, uint32_t threads, bool barrier, hsa_signal_t done, bool ring = true) {
        uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
        hsa_kernel_dispatch_packet_t b = {};
        b.workgroup_size_x = (uint16_t)threads; b.workgroup_size_y = 1; b.workgroup_size_z = 1;
        b.grid_size_x = wgs * threads; b.grid_size_y = 1; b.grid_size_z = 1;
        b.private_segment_size = priv_; b.group_segment_size = group; b.kernel_object = kobj; b.kernarg_address = (void*)kernarg;
        b.completion_signal = done;
        memcpy((char*)p + 4, (char*)&b + 4, sizeof(b) - 4);
        const int fence = (!barrier && g_nofence) ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT;
        uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                          (fence << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (fence << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n(&p->full_header, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
        last = idx;
        if (ring) hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    }
    void ring() { hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)last); }
    uint64_t last = 0;
};

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: aql_lab <code object>\n"); return 2; }
    g_nofence = argc > 2 && !strcmp(argv[2], "nofence");
    if (g_nofence) printf("packets without the barrier bit carry NO acquire / release fence\n");
    CK(hipSetDevice(0)); CK(hipFree(0));
    HK(hsa_init());
    HK(hsa_iterate_agents(pick_gpu, nullptr));
    if (!g_have) { printf("no GPU agent\n"); return 1; }
    char name[64] = {}; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name); printf("agent %s\n", name);

    std::ifstream f(argv[1], std::ios::binary); std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (co.empty()) { printf("cannot read %s\n", argv[1]); return 1; }
    hsa_code_object_reader_t rd; HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &rd));
    hsa_executable_t ex; HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    HK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    HK(hsa_executable_freeze(ex, nullptr));
    hsa_executable_symbol_t sym; HK(hsa_executable_get_symbol_by_name(ex, "lab_kernel.kd", &g_gpu, &sym));
    Queue Q = {};
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &Q.kobj));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &Q.group));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &Q.priv_));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &Q.kasz));
    printf("kernel object %#llx group %u private %u kernarg %u (explicit %zu)\n", (unsigned long long)Q.kobj, Q.group, Q.priv_, Q.kasz, sizeof(Args));
    HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &Q.q));
    hsa_signal_t done; HK(hsa_signal_create(1, 0, nullptr, &done));
    hsa_signal_t doneA; HK(hsa_signal_create(1, 0, nullptr, &doneA));      // without the barrier bit B may complete BEFORE A: wait for both

    const int NMAX = 16384;
    unsigned long long *d_out, *d_seen; unsigned *d_flag, *d_data, *d_bad; char* d_ka;
    CK(hipMalloc(&d_out, 16 * NMAX)); CK(hipMalloc(&d_seen, 16 * NMAX)); CK(hipMalloc(&d_flag, 4096)); CK(hipMalloc(&d_data, 4 * NMAX)); CK(hipMalloc(&d_bad, 4));
    CK(hipMemset(d_flag, 0, 4096)); CK(hipMemset(d_bad, 0, 4)); CK(hipMemset(d_data, 0, 4 * NMAX));
    const int KA = 512, NKA = 2048;
    CK(hipMalloc(&d_ka, KA * NKA));
    std::vector<char> h_ka(KA * NKA, 0);
    std::vector<unsigned long long> h(2 * NMAX), hs(2 * NMAX);
    bool useA = false;
    auto wait_done = [&]() {
        hsa_signal_value_t v = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
        if (v >= 1) { printf("TIMEOUT waiting for the queue\n"); exit(3); }
        hsa_signal_store_relaxed(done, 1);
        if (useA) {
            v = hsa_signal_wait_scacquire(doneA, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
            if (v >= 1) { printf("TIMEOUT waiting for the queue (A)\n"); exit(3); }
            hsa_signal_store_relaxed(doneA, 1);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, 16 * NMAX, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), d_seen, 16 * NMAX, hipMemcpyDeviceToHost));
    };
    auto setarg = [&](int slot, const Args& a) { memcpy(h_ka.data() + slot * KA, &a, sizeof(a)); };
    auto push_args = [&]() { CK(hipMemcpy(d_ka, h_ka.data(), h_ka.size(), hipMemcpyHostToDevice)); CK(hipDeviceSynchronize()); };
    hsa_signal_t none = {0};

    // (1) + (2): overlap and dispatch order of two packets
    for (int geo = 0; geo < 2; ++geo) {
        int nA = geo ? 4096 : 1, nB = geo ? 256 : 1; long long tA = geo ? 500 : 5000;
        for (int barrier = 1; barrier >= 0; --barrier) {
            Args a = {}; a.out = d_out; a.base = 0; a.ticks = tA; setarg(0, a);
            Args b = {}; b.out = d_out; b.base = nA; b.ticks = 100; setarg(1, b);
            push_args();
            for (int rep = 0; rep < 2; ++rep) {
                useA = true;
                Q.submit(d_ka, nA, 256, true, doneA, false);
                Q.submit(d_ka + KA, nB, 256, barrier, done, true);
                wait_done();
                useA = false;
            }
            unsigned long long a0 = ~0ull, a0max = 0, a1 = 0, b0 = ~0ull, b1 = 0;
            for (int i = 0; i < nA; ++i) { a0 = std::min(a0, h[2 * i]); a0max = std::max(a0max, h[2 * i]); a1 = std::max(a1, h[2 * i + 1]); }
            for (int i = nA; i < nA + nB; ++i) { b0 = std::min(b0, h[2 * i]); b1 = std::max(b1, h[2 * i + 1]); }
            printf("A=%4d wg x %5.1f us, B=%3d wg, barrier bit %d: A last start %+6.2f end %+6.2f | B first start %+6.2f end %+6.2f us => %s, B starts %s the last A workgroup\n",
                   nA, tA / 100.0, nB, barrier, (a0max - a0) / 100.0, (a1 - a0) / 100.0, ((long long)b0 - (long long)a0) / 100.0, ((long long)b1 - (long long)a0) / 100.0,
                   b0 < a1 ? "OVERLAP" : "serial", b0 >= a0max ? "after" : "BEFORE");
        }
    }
    // (4) a chain of 512 one-workgroup... and 256-workgroup empty kernels, with and without the barrier bit
    for (int wgs : {1, 256, 688}) for (int barrier = 1; barrier >= 0; --barrier) {
        const int N = 512;
        for (int i = 0; i < N; ++i) { Args a = {}; a.out = d_out; a.base = 0; a.ticks = 0; setarg(i, a); }
        push_args();
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < N; ++i) { Args a = {}; a.out = d_out; a.base = (i == 0 || i == N - 1) ? (i ? 1 : 0) * wgs : 2 * wgs; a.ticks = 0; setarg(i, a); }
            push_args();
            for (int i = 0; i < N; ++i) Q.submit(d_ka + i * KA, wgs, 256, (i == 0 || i == N - 1) ? true : barrier, i == N - 1 ? done : none, false);
            Q.ring(); wait_done();
            unsigned long long s = ~0ull, e = 0;
            for (int i = 0; i < wgs; ++i) { s = std::min(s, h[2 * i]); e = std::max(e, h[2 * (wgs + i) + 1]); }
            best = std::min(best, (double)(e - s) / 100.0 / (N - 1));
        }
        printf("chain of %d empty kernels x %3d workgroups, barrier bit %d: %.2f us per kernel\n", N, wgs, barrier, best);
    }
    // (3) dependency through a release/acquire counter: A (nA workgroups, ~5 us each) signals, B (no barrier bit) waits, then reads A's data
    for (int nA : {256, 688}) for (int nB : {256, 688}) {
        unsigned epoch = 0; std::vector<double> lat_seen, lat_data, b_wait;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemset(d_flag, 0, 4096)); CK(hipDeviceSynchronize());
            unsigned stamp = 1000 + rep + nA + nB;
            Args a = {}; a.out = d_out; a.base = 0; a.ticks = 500; a.sig = d_flag; a.data = d_data; a.ndata = nA; a.stamp = stamp; setarg(0, a);
            Args b = {}; b.out = d_out; b.base = nA; b.ticks = 0; b.wait = d_flag; b.wait_val = nA; b.data = d_data; b.ndata = nA; b.stamp = stamp; b.seen = d_seen; b.bad = d_bad; setarg(1, b);
            push_args();
            useA = true;
            Q.submit(d_ka, nA, 256, true, doneA, false);
            Q.submit(d_ka + KA, nB, 256, false, done, true);
            wait_done();
            useA = false;
            unsigned long long a1 = 0, b0 = ~0ull, smin = ~0ull, smax = 0, dmax = 0;
            for (int i = 0; i < nA; ++i) a1 = std::max(a1, h[2 * i + 1]);
            for (int i = nA; i < nA + nB; ++i) { b0 = std::min(b0, h[2 * i]); smin = std::min(smin, hs[2 * i]); smax = std::max(smax, hs[2 * i]); dmax = std::max(dmax, hs[2 * i + 1]); }
            if (rep) { lat_seen.push_back(((long long)smax - (long long)a1) / 100.0); lat_data.push_back(((long long)dmax - (long long)a1) / 100.0); b_wait.push_back(((long long)a1 - (long long)b0) / 100.0); }
            (void)epoch; (void)smin;
        }
        std::sort(lat_seen.begin(), lat_seen.end()); std::sort(lat_data.begin(), lat_data.end()); std::sort(b_wait.begin(), b_wait.end());
        unsigned bad = 0; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("dependency A=%3d -> B=%3d workgroups: last A workgroup's signal -> last B workgroup sees the counter %.2f us (median of 5; min %.2f), has A's data %.2f us (min %.2f); first B started %.2f us before A's end; bad reads %u\n",
               nA, nB, lat_seen[lat_seen.size() / 2], lat_seen[0], lat_data[lat_data.size() / 2], lat_data[0], b_wait[b_wait.size() / 2], bad);
    }
    hsa_queue_destroy(Q.q);
    return 0;
}
