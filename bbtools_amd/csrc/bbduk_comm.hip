// bbduk_comm.hip -- the multi-GPU surface of include/bbduk_gpu.h (SURVEY.md §8b "allreduce_counters", §8e).
//
// The path shards embarrassingly: whole pairs go to 1/2/4/8 GPUs, the k-mer map is replicated, and the ONLY exchange is
// one sum of the int64 counter vector at the end of a run -- what BBDukProcessorS.add does when the per-thread
// processors are merged (bbduk/BBDukProcessorS.java:300-342).  Here that sum is one ncclAllReduce(ncclInt64, ncclSum)
// (RCCL, xGMI); the vector is 16 + 2*numScaffolds values (<= ~3 KB), so the collective is latency-bound and nothing is
// bucketed or overlapped.
//
// Two ways to form the group, matching the two kinds of host:
//   * one PROCESS per GPU (bench.py, torchrun-style launchers): rank 0 calls bbduk_comm_unique_id, the launcher hands the
//     128 bytes to every rank, every rank calls bbduk_comm_create(h, nranks, rank, id);
//   * one process driving SEVERAL GPUs (the JVM host, bbduk_cli devices=0,1,..): bbduk_comm_create_local(handles, n).
//     Handles that share a device are summed on that device first (a 1-block kernel), the device leaders run the RCCL
//     all-reduce, the result is copied back to the followers: RCCL refuses two ranks on one device.
//
// librccl is opened with dlopen at the first comm call, so that a single-GPU host needs no RCCL at all; in a process
// that already holds a librccl.so.1 (PyTorch ships one) the same copy is used.
#include <dlfcn.h>
#include <string.h>
#include <map>
#include <thread>
#include "bbduk_internal.h"
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {                                  // loaded once; nullptr-lib + err on failure
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.lib) break;
        }
        if (!R.lib) { R.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(R.lib, n); if (!p) { ok = false; R.err = std::string("librccl lacks ") + n; } return p; };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(sym("ncclCommInitAll"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
        R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { dlclose(R.lib); R.lib = nullptr; }
    });
    return &R;
}

int nccl_fail(bbduk_handle* h, const char* what, ncclResult_t r) {
    if (h) h->err = std::string(what) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r) : "rccl error");
    return BBDUK_ERR_DEVICE;
}

// dst[i] += src[i]; the vectors are a few hundred values: one block
__global__ void bbduk_add_counters_kernel(int64_t* __restrict__ dst, const int64_t* __restrict__ src, const int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] += src[i];
}

}  // namespace

// One per process-rank (multi-process) or one shared by the handles of a local group.
struct bbduk_comm {
    // multi-process form
    ncclComm_t comm = nullptr; int nranks = 0, rank = 0;
    // local form: every member handle points at the same object
    bool local = false;
    std::vector<bbduk_handle*> members;          // in the caller's order
    std::vector<int> leaderOf;                   // members[i]'s device leader (index into members)
    std::vector<int> leaders;                    // one member per distinct device
    std::vector<ncclComm_t> comms;               // comms[j] belongs to leaders[j]
    std::vector<int64_t*> scratch;               // scratch[j]: a counter vector on leaders[j]'s device (receives a follower's copy)
    int ncounters = 0;
    std::mutex mu;
};

// Start loading the collective library now, on a thread of its own: librccl.so.1 is several hundred megabytes and its load registers every code object
// with the HIP runtime -- tens of seconds from a cold page cache.  A host that will form a group over several devices calls this first thing, so that
// the load overlaps its table build; the comm calls then find the library loaded (or wait for the rest of the load).  Never needed for correctness.
extern "C" int bbduk_comm_preload(void) {
    std::thread([]() { rccl(); }).detach();
    return BBDUK_OK;
}

extern "C" int bbduk_comm_unique_id(uint8_t* id128) {
    if (!id128) return BBDUK_ERR_ARG;
    Rccl* R = rccl();
    if (!R->lib) return BBDUK_ERR_DEVICE;
    static_assert(sizeof(ncclUniqueId) == BBDUK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return BBDUK_ERR_DEVICE;
    memcpy(id128, &id, sizeof id);
    return BBDUK_OK;
}

extern "C" int bbduk_comm_create(bbduk_handle* h, int32_t nranks, int32_t rank, const uint8_t* id128) {
    if (!h) return BBDUK_ERR_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks || !id128) return fail(h, BBDUK_ERR_ARG, "comm_create: bad rank / nranks / id");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->comm) return fail(h, BBDUK_ERR_STATE, "the handle already belongs to a communicator");
    Rccl* R = rccl();
    if (!R->lib) { h->err = R->err; return BBDUK_ERR_DEVICE; }
    HIP_TRY(h, hipSetDevice(h->p.device));
    bbduk_comm* c = new (std::nothrow) bbduk_comm();
    if (!c) return BBDUK_ERR_NOMEM;
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    const ncclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail(h, "ncclCommInitRank", r); }
    c->nranks = nranks; c->rank = rank; c->ncounters = bbduk_counters_len(h);
    h->comm = c;
    return BBDUK_OK;
}

extern "C" int bbduk_comm_create_local(bbduk_handle** hs, int32_t n) {
    if (!hs || n < 1) return BBDUK_ERR_ARG;
    for (int i = 0; i < n; i++) if (!hs[i]) return BBDUK_ERR_ARG;
    bbduk_handle* h0 = hs[0];
    const int nc = bbduk_counters_len(h0);
    for (int i = 0; i < n; i++) {
        if (hs[i]->comm) return fail(h0, BBDUK_ERR_STATE, "a handle already belongs to a communicator");
        if (bbduk_counters_len(hs[i]) != nc) return fail(h0, BBDUK_ERR_ARG, "the handles of a group must have the same numScaffolds");
        for (int j = 0; j < i; j++) if (hs[j] == hs[i]) return fail(h0, BBDUK_ERR_ARG, "the same handle twice");
    }
    bbduk_comm* c = new (std::nothrow) bbduk_comm();
    if (!c) return BBDUK_ERR_NOMEM;
    c->local = true; c->ncounters = nc;
    c->members.assign(hs, hs + n);
    c->leaderOf.resize(n);
    std::map<int, int> firstOnDevice;
    std::vector<int> devs;
    for (int i = 0; i < n; i++) {
        auto it = firstOnDevice.find(hs[i]->p.device);
        if (it == firstOnDevice.end()) { firstOnDevice[hs[i]->p.device] = i; c->leaders.push_back(i); devs.push_back(hs[i]->p.device); c->leaderOf[i] = i; }
        else c->leaderOf[i] = it->second;
    }
    c->comms.assign(c->leaders.size(), nullptr);
    // A group on ONE device needs no collective library at all: its handles are summed on that device (step 1 of bbduk_allreduce_counters_local) and that
    // is the result.  Round 5: this is what `bbduk_cli devices=0,0,0` had been stalling in -- dlopen of a 573 MB librccl.so.1 whose code objects are all
    // registered with the HIP runtime at load; on a fresh box that is a cold read of the whole file (45-130 s measured, wchan folio_wait_bit_common:
    // profiles/r05_hang_hunt.txt), which a 120 s test timeout read as a hang.  With several devices RCCL is needed: bbduk_comm_preload() starts that
    // load early so that it overlaps the table build.
    Rccl* R = nullptr;
    if (devs.size() > 1) {
        R = rccl();
        if (!R->lib) { h0->err = R->err; delete c; return BBDUK_ERR_DEVICE; }
        const ncclResult_t r = R->CommInitAll(c->comms.data(), (int)devs.size(), devs.data());
        if (r != ncclSuccess) { delete c; return nccl_fail(h0, "ncclCommInitAll", r); }
    }
    c->scratch.assign(c->leaders.size(), nullptr);
    for (size_t j = 0; j < c->leaders.size(); j++) {
        if (hipSetDevice(devs[j]) != hipSuccess || hipMalloc(&c->scratch[j], (size_t)nc * sizeof(int64_t)) != hipSuccess) {
            for (size_t q = 0; q < c->leaders.size(); q++) { if (c->scratch[q]) { hipSetDevice(devs[q]); hipFree(c->scratch[q]); } if (R && c->comms[q]) R->CommDestroy(c->comms[q]); }
            delete c;
            return fail(h0, BBDUK_ERR_NOMEM, "hipMalloc (communicator scratch)");
        }
    }
    c->nranks = (int)devs.size();
    for (int i = 0; i < n; i++) hs[i]->comm = c;
    return BBDUK_OK;
}

extern "C" int bbduk_comm_destroy(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    bbduk_comm* c = h->comm;
    if (!c) return BBDUK_OK;
    if (c->local) {                                                // the first member to leave tears the group down for all of them
        Rccl* R = c->leaders.size() > 1 ? rccl() : nullptr;
        for (bbduk_handle* m : c->members) m->comm = nullptr;
        for (size_t j = 0; j < c->leaders.size(); j++) {
            hipSetDevice(c->members[c->leaders[j]]->p.device);
            hipFree(c->scratch[j]);
            if (R && c->comms[j]) R->CommDestroy(c->comms[j]);
        }
    } else {
        hipSetDevice(h->p.device);
        if (c->comm) rccl()->CommDestroy(c->comm);
        h->comm = nullptr;
    }
    delete c;
    return BBDUK_OK;
}

extern "C" int bbduk_comm_size(const bbduk_handle* h) { return (h && h->comm) ? h->comm->nranks : 0; }

extern "C" int bbduk_allreduce_counters_device(bbduk_handle* h, int64_t* d_counters, void* stream) {
    if (!h || !d_counters) return fail(h, BBDUK_ERR_ARG, "allreduce_counters: null argument");
    bbduk_comm* c = h->comm;
    if (!c || c->local) return fail(h, BBDUK_ERR_STATE, "allreduce_counters: the handle is not a rank of a multi-process communicator (bbduk_comm_create)");
    HIP_TRY(h, hipSetDevice(h->p.device));
    const ncclResult_t r = rccl()->AllReduce(d_counters, d_counters, (size_t)c->ncounters, ncclInt64, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail(h, "ncclAllReduce", r);
    return BBDUK_OK;
}

// the same all-reduce over another int64 vector of this handle's rank (the counter vector of a seal_handle: seal_allreduce_counters)
int bbduk_comm_allreduce_i64(bbduk_handle* h, int64_t* d_buf, int64_t n, void* stream) {
    if (!h || !d_buf || n < 1) return fail(h, BBDUK_ERR_ARG, "allreduce: null argument");
    bbduk_comm* c = h->comm;
    if (!c || c->local) return fail(h, BBDUK_ERR_STATE, "allreduce: the handle is not a rank of a multi-process communicator (bbduk_comm_create / seal_comm_create)");
    HIP_TRY(h, hipSetDevice(h->p.device));
    const ncclResult_t r = rccl()->AllReduce(d_buf, d_buf, (size_t)n, ncclInt64, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail(h, "ncclAllReduce", r);
    return BBDUK_OK;
}

extern "C" int bbduk_allreduce_counters(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    if (h->comm && h->comm->local) return bbduk_allreduce_counters_local(h->comm->members.data(), (int32_t)h->comm->members.size());
    std::lock_guard<std::mutex> g(h->mu);
    const int rc = bbduk_allreduce_counters_device(h, h->d_counters, h->stream);
    if (rc != BBDUK_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return BBDUK_OK;
}

extern "C" int bbduk_allreduce_counters_local(bbduk_handle** hs, int32_t n) {
    if (!hs || n < 1 || !hs[0]) return BBDUK_ERR_ARG;
    bbduk_handle* h0 = hs[0];
    bbduk_comm* c = h0->comm;
    if (!c || !c->local) return fail(h0, BBDUK_ERR_STATE, "allreduce_counters_local: no local communicator (bbduk_comm_create_local)");
    if ((size_t)n != c->members.size()) return fail(h0, BBDUK_ERR_ARG, "allreduce_counters_local: pass the whole group");
    for (int i = 0; i < n; i++) if (hs[i] != c->members[i]) return fail(h0, BBDUK_ERR_ARG, "allreduce_counters_local: pass the group in the order given to bbduk_comm_create_local");
    std::lock_guard<std::mutex> g(c->mu);
    Rccl* R = c->leaders.size() > 1 ? rccl() : nullptr;            // (one device: no collective, see bbduk_comm_create_local)
    const size_t bytes = (size_t)c->ncounters * sizeof(int64_t);
    // every member's own stream must have finished what it was accumulating
    for (bbduk_handle* m : c->members) { HIP_TRY(h0, hipSetDevice(m->p.device)); HIP_TRY(h0, hipStreamSynchronize(m->stream)); }
    // 1. followers -> their device leader (same device: a copy into the leader's scratch + one add, on the leader's stream)
    for (int i = 0; i < n; i++) {
        const int l = c->leaderOf[i];
        if (l == i) continue;
        bbduk_handle* L = c->members[l];
        size_t j = 0; while (c->leaders[j] != l) j++;
        HIP_TRY(h0, hipSetDevice(L->p.device));
        HIP_TRY(h0, hipMemcpyAsync(c->scratch[j], c->members[i]->d_counters, bytes, hipMemcpyDeviceToDevice, L->stream));
        bbduk_add_counters_kernel<<<dim3(1), dim3(256), 0, L->stream>>>(L->d_counters, c->scratch[j], c->ncounters);
        HIP_TRY(h0, hipGetLastError());
    }
    // 2. leaders: one grouped RCCL all-reduce (sum, int64) over the distinct devices
    if (R) {
        ncclResult_t r = R->GroupStart();
        if (r != ncclSuccess) return nccl_fail(h0, "ncclGroupStart", r);
        for (size_t j = 0; j < c->leaders.size(); j++) {
            bbduk_handle* L = c->members[c->leaders[j]];
            hipSetDevice(L->p.device);
            r = R->AllReduce(L->d_counters, L->d_counters, (size_t)c->ncounters, ncclInt64, ncclSum, c->comms[j], L->stream);
            if (r != ncclSuccess) { R->GroupEnd(); return nccl_fail(h0, "ncclAllReduce", r); }
        }
        r = R->GroupEnd();
        if (r != ncclSuccess) return nccl_fail(h0, "ncclGroupEnd", r);
    }
    // 3. back to the followers
    for (int i = 0; i < n; i++) {
        const int l = c->leaderOf[i];
        if (l == i) continue;
        bbduk_handle* L = c->members[l];
        HIP_TRY(h0, hipSetDevice(L->p.device));
        HIP_TRY(h0, hipMemcpyAsync(c->members[i]->d_counters, L->d_counters, bytes, hipMemcpyDeviceToDevice, L->stream));
    }
    for (size_t j = 0; j < c->leaders.size(); j++) {
        bbduk_handle* L = c->members[c->leaders[j]];
        HIP_TRY(h0, hipSetDevice(L->p.device));
        HIP_TRY(h0, hipStreamSynchronize(L->stream));
    }
    return BBDUK_OK;
}
