// The ONE collective of the path: independent units are sharded over ranks with no data-path
// communication; at the end every rank's device-resident result block is gathered over RCCL (xGMI inside a node).
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy the process has already loaded -- e.g. the one PyTorch ships --
// or the ROCm one), so libgprx.so itself has no link-time dependency on it and single-GPU users never load it.
// One communicator = one device, one private HIP stream; collectives are asynchronous on that stream and take DEVICE
// pointers (no host bounce).  Gathering to a root uses grouped ncclSend / ncclRecv so that all inbound xGMI links of the
// root carry traffic at once (7 links x ~153 GB/s: a ring all-gather would be bound by one link per hop).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

namespace gprx {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // (optional: what RCCL itself says about the communicator)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  std::string error;

  std::mutex mu;

  bool load() {
    std::lock_guard<std::mutex> guard(mu);
    if (lib) return true;
    error.clear();
    const bool dbg = getenv("GPRX_COMM_DEBUG") != nullptr;
    auto try_open = [&](const std::string& name, int flags, const char* stage) {
      dlerror();
      void* h = dlopen(name.c_str(), flags);
      if (dbg) fprintf(stderr, "[gprx comm] %s dlopen(%s) -> %s%s\n", stage, name.c_str(), h ? "ok" : "failed: ", h ? "" : (dlerror() ? dlerror() : "?"));
      return h;
    };
    const char* names[] = {"librccl.so.1", "librccl.so"};
    {
      // (1) the RCCL that sits NEXT TO the HIP runtime this library calls, by full path and RTLD_LOCAL -- even when another
      // RCCL is already in the process.  Measured on the MI355X box: every process there has PyTorch's librccl.so (with
      // PyTorch's own libamdhip64 / libhsa-runtime64) mapped before user code runs; using THAT copy from a library bound to
      // /opt/rocm's HIP fails in ncclCommInitRank ("pfn_hsa_system_get_info failed with 4107", "no ROCm-capable device"):
      // its HIP / HSA runtime is a second, uninitialised one.  One ROCm tree for HIP, HSA and RCCL.
      hipError_t (*volatile fn)(unsigned int) = &hipInit;  // (volatile: the function's own address, not this library's PLT stub)
      Dl_info info;
      if (dladdr(reinterpret_cast<void*>(fn), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        if (dbg) fprintf(stderr, "[gprx comm] HIP runtime in use: %s\n", dir.c_str());
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos && dir.find("libamdhip64") != std::string::npos) {
          dir.resize(slash + 1);
          for (const char* n : names)
            if ((lib = try_open(dir + n, RTLD_NOW | RTLD_LOCAL, "next to HIP:"))) break;
        }
      }
    }
    // (2) an RCCL the process has already loaded, (3) the search path
    if (!lib)
      for (const char* n : names)
        if ((lib = try_open(n, RTLD_NOW | RTLD_NOLOAD, "already loaded?"))) break;
    if (!lib)
      for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((lib = try_open(n, RTLD_NOW | RTLD_LOCAL, "search path:"))) break;
    if (!lib) {
      error = std::string("cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
      return false;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(lib, name);
      if (!p && error.empty()) error = std::string("RCCL symbol missing: ") + name;
      return p;
    };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
    CommUserRank = reinterpret_cast<decltype(CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    if (!error.empty()) {
      lib = nullptr;
      return false;
    }
    return true;
  }
};

inline RcclApi& rccl() {
  static RcclApi api;
  return api;
}

}  // namespace gprx
