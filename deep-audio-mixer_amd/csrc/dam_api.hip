// dam_api.hip -- library identification entry points of include/dam_hip.h, and the host-side fork guard.
#include "dam_common.h"

#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>

extern "C" const char* dam_arch(void) { return "gfx950"; }
extern "C" int dam_abi_version(void) { return DAM_ABI_VERSION; }

namespace {

// ---- what does the GPU stack know about a host address?
// ROCr's own view (hsa_amd_pointer_info) gives the exact extent of the allocation behind an address (the kernel merges
// neighbouring allocations into one mapping); hipPointerGetAttributes tells whether it is a page-locked buffer the HIP layer
// handed out.  The symbol is taken from the libhsa-runtime64 that is ALREADY loaded (PyTorch wheels ship their
// own copy: loading a second instance would answer "not initialised" for everything).
enum { PTR_UNKNOWN = 0, PTR_HSA = 1 };                 // hsa_amd_pointer_type_t: allocated by an HSA memory allocator
struct PointerInfo {                                    // hsa_amd_pointer_info_t (hsa_ext_amd.h); "can only grow"
    uint32_t size;
    int32_t type;
    void* agentBaseAddress;
    void* hostBaseAddress;
    size_t sizeInBytes;
    void* userData;
    uint64_t agentOwner;
    uint8_t global_flags;
    uint8_t registered;
    uint8_t pad[46];                                    // room for fields of newer runtimes
};
typedef int (*pointer_info_fn)(const void*, PointerInfo*, void* (*)(size_t), uint32_t*, void**);

int find_hsa(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && strstr(info->dlpi_name, "libhsa-runtime64")) {
        strncpy(static_cast<char*>(out), info->dlpi_name, 1023);
        return 1;
    }
    return 0;
}

pointer_info_fn hsa_pointer_info() {
    static pointer_info_fn fn = [] {
        char path[1024] = {0};
        if (!dl_iterate_phdr(find_hsa, path)) return (pointer_info_fn) nullptr;
        void* h = dlopen(path, RTLD_NOLOAD | RTLD_LAZY);
        return h ? reinterpret_cast<pointer_info_fn>(dlsym(h, "hsa_amd_pointer_info")) : (pointer_info_fn) nullptr;
    }();
    return fn;
}

// The HSA allocation that contains p: [base, base + bytes) in host addresses; false if p is ordinary memory (or memory that
// was merely LOCKED for a transfer: that is the program's own, a child may need it).
bool hsa_allocation(pointer_info_fn fn, const void* p, unsigned long* base, unsigned long* bytes) {
    PointerInfo pi;
    memset(&pi, 0, sizeof(pi));
    pi.size = sizeof(pi);
    if (fn(p, &pi, nullptr, nullptr, nullptr) != 0 || pi.type != PTR_HSA || !pi.hostBaseAddress || !pi.sizeInBytes) return false;
    *base = (unsigned long)pi.hostBaseAddress;
    *bytes = (unsigned long)pi.sizeInBytes;
    return true;
}

// Fallback without ROCr's symbol: page-locked memory the HIP layer knows (hipHostMalloc / hipHostRegister).
bool is_hip_host_page(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();          // ordinary memory on runtimes that report it as an error: clear the sticky status
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

}  // namespace

// see include/dam_hip.h
extern "C" int dam_host_dontfork_pinned(int64_t* n_mappings_host, int64_t* n_bytes_host) {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return DAM_ERR_UNSUPPORTED;
    // the mappings are collected first: marking one splits / merges VMAs and would move the file under the reader
    struct Range { unsigned long lo, hi; };
    static const int MAX_RANGES = 16384;
    Range* r = new Range[MAX_RANGES];
    int n = 0;
    char line[1024];
    while (n < MAX_RANGES && fgets(line, sizeof(line), f)) {
        unsigned long lo = 0, hi = 0, off = 0, ino = 0;
        char perms[8] = {0}, dev[16] = {0};
        int consumed = 0;
        if (sscanf(line, "%lx-%lx %7s %lx %15s %lu %n", &lo, &hi, perms, &off, dev, &ino, &consumed) < 6) continue;
        // candidates: private, readable + writable, anonymous (no inode, no name such as [heap] / [stack])
        if (perms[0] != 'r' || perms[1] != 'w' || perms[3] != 'p' || ino != 0) continue;
        const char* name = line + consumed;
        if (*name != '\0' && *name != '\n') continue;
        r[n].lo = lo;
        r[n].hi = hi;
        ++n;
    }
    fclose(f);
    int64_t marked = 0, bytes = 0;
    const unsigned long page = 4096;
    const pointer_info_fn info = hsa_pointer_info();
    for (int i = 0; i < n; ++i) {
        const unsigned long lo = r[i].lo, hi = r[i].hi;
        if (info) {
            // ROCr carves its host allocations out of an address range it reserved: what sits next to one inside a mapping is
            // another one (the kernel merges neighbours with equal flags).  Walk the mapping allocation by allocation from its
            // start and stop at the first address ROCr does not own -- ordinary memory is never marked.
            unsigned long p = lo;
            while (p < hi) {
                unsigned long base = 0, len = 0;
                if (!hsa_allocation(info, (const void*)p, &base, &len) || base + len <= p) break;
                // ... and only what the HIP layer handed out as page-locked DATA buffers (hipHostMalloc: torch's pinned
                // blocks).  The runtime's own pools (signals, kernel arguments, staging) are skipped on purpose: a forked
                // child that runs a destructor of an inherited GPU object calls into its copy of the runtime, which reads
                // them -- unmapped, that call would fault (seen: a DataLoader worker killed by SIGSEGV) instead of
                // returning an error.  No destructor reads the CONTENTS of a pinned data buffer.
                if (!is_hip_host_page((const void*)p)) {
                    p = (base + len + page - 1) & ~(page - 1);
                    continue;
                }
                unsigned long a = (base > lo ? base : lo) & ~(page - 1), b = base + len < hi ? base + len : hi;
                b = (b + page - 1) & ~(page - 1);
                if (b > hi) b = hi;
                if (a < b && madvise((void*)a, b - a, MADV_DONTFORK) == 0) {
                    ++marked;
                    bytes += (int64_t)(b - a);
                }
                p = b;
            }
            continue;
        }
        // a mapping counts only if its first, middle and last page are all page-locked host memory
        const unsigned long mid = lo + (((hi - lo) / 2) & ~(page - 1));
        if (!is_hip_host_page((const void*)lo) || !is_hip_host_page((const void*)mid) || !is_hip_host_page((const void*)(hi - page)))
            continue;
        if (madvise((void*)lo, hi - lo, MADV_DONTFORK) == 0) {
            ++marked;
            bytes += (int64_t)(hi - lo);
        }
    }
    delete[] r;
    if (n_mappings_host) *n_mappings_host = marked;
    if (n_bytes_host) *n_bytes_host = bytes;
    return DAM_OK;
}

// ---- step marks (include/dam_hip.h, ABI 15): an event another stream can wait for, recorded INSIDE a captured step
extern "C" int dam_step_mark_create(void** mark_host) {
    if (!mark_host) return DAM_ERR_BAD_ARG;
    hipEvent_t ev = nullptr;
    // no system-scope fence: a mark orders streams of ONE device (kernel boundaries and copy completion carry the data); the
    // default event's cache write-back + invalidate in front of every step cost the following kernels 0.1 ms per step
    // (tools/sync_cost_probe.py, profiles/r05_sync_cost_probe.txt)
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    *mark_host = ev;
    return DAM_OK;
}

extern "C" int dam_step_mark_record(void* mark, void* stream) {
    if (!mark) return DAM_ERR_BAD_ARG;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    // capturing: an event-record NODE that every replay executes; the event stays usable from streams outside the graph
    hipError_t e;
    if (st != hipStreamCaptureStatusActive) {
        e = hipEventRecord((hipEvent_t)mark, (hipStream_t)stream);
    } else {
        e = hipEventRecordWithFlags((hipEvent_t)mark, (hipStream_t)stream, hipEventRecordExternal);
        if (e != hipSuccess) {
            // the runtime torch ships answers "invalid argument" here: put the event-record node into the graph under capture by
            // hand -- behind everything the stream has captured so far, and everything captured later behind it
            (void)hipGetLastError();
            hipGraph_t graph = nullptr;
            const hipGraphNode_t* deps = nullptr;
            size_t ndeps = 0;
            hipGraphNode_t node = nullptr;
            e = hipStreamGetCaptureInfo_v2((hipStream_t)stream, &st, nullptr, &graph, &deps, &ndeps);
            if (e == hipSuccess) e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)mark);
            if (e == hipSuccess) e = hipStreamUpdateCaptureDependencies((hipStream_t)stream, &node, 1, hipStreamSetCaptureDependencies);
        }
    }
    if (e != hipSuccess) {
        fprintf(stderr, "dam_step_mark_record: %s (capturing %d)\n", hipGetErrorString(e), (int)(st == hipStreamCaptureStatusActive));
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    return DAM_OK;
}

extern "C" int dam_step_mark_wait(void* mark, void* stream) {
    if (!mark) return DAM_ERR_BAD_ARG;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    if (st != hipStreamCaptureStatusNone) return DAM_ERR_BAD_ARG;      // the waiter is a copy stream outside every graph
    if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)mark, 0) != hipSuccess) {
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    return DAM_OK;
}

extern "C" int dam_step_mark_synchronize(void* mark) {
    if (!mark) return DAM_ERR_BAD_ARG;
    if (hipEventSynchronize((hipEvent_t)mark) != hipSuccess) {
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    return DAM_OK;
}

extern "C" int dam_step_mark_destroy(void* mark) {
    if (!mark) return DAM_ERR_BAD_ARG;
    if (hipEventDestroy((hipEvent_t)mark) != hipSuccess) {
        (void)hipGetLastError();
        return DAM_ERR_LAUNCH;
    }
    return DAM_OK;
}
