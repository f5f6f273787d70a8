// runtime_identity.h — which HIP / HSA / RCCL shared objects are mapped into this process, and which of them serves this library.
//
// Why this exists (root-caused in round 4): PyTorch's ROCm wheel ships private copies of libamdhip64.so, libhsa-runtime64.so
// and librccl.so under torch/lib (RPATH $ORIGIN).  They carry the SAME sonames as /opt/rocm's (libamdhip64.so.7, libhsa-runtime64.so.1)
// but torch links them by their UNVERSIONED file names, so what the dynamic loader does depends on the order of arrival:
//   torch first      libptamd.so's DT_NEEDED "libamdhip64.so.7" matches the soname of torch's already-loaded copy: ONE runtime.
//   libptamd first   /opt/rocm's copy is mapped (RUNPATH); torch's DT_NEEDED "libamdhip64.so" matches no loaded soname, is found through
//                    $ORIGIN as a different file and mapped as well: TWO HIP runtimes over TWO HSA runtimes.  The one initialised second
//                    cannot acquire the process's GPU VM (the kernel driver binds it to the first runtime's DRM file) and reports
//                    "no HIP GPUs".
// A process must hold exactly one HIP runtime.  pt_create refuses to start otherwise, and says which objects collide.
#pragma once
#include <dlfcn.h>
#include <link.h>

#include <cstring>
#include <string>
#include <vector>

namespace pt {

struct RuntimeObjects {
  std::vector<std::string> hip, hsa, rccl;
};

inline int runtime_objects_cb(struct dl_phdr_info* info, size_t, void* data) {
  auto* o = static_cast<RuntimeObjects*>(data);
  const char* name = info->dlpi_name;
  if (!name || !*name) return 0;
  const char* base = strrchr(name, '/');
  base = base ? base + 1 : name;
  auto add = [&](std::vector<std::string>& v) {
    for (const std::string& s : v) if (s == name) return;
    v.push_back(name);
  };
  if (strncmp(base, "libamdhip64.so", 14) == 0) add(o->hip);
  else if (strncmp(base, "libhsa-runtime64.so", 19) == 0) add(o->hsa);
  else if (strncmp(base, "librccl.so", 10) == 0) add(o->rccl);
  return 0;
}

inline RuntimeObjects mapped_runtime_objects() {
  RuntimeObjects o;
  dl_iterate_phdr(runtime_objects_cb, &o);
  return o;
}

// the shared object a function pointer (as THIS library resolved it) lives in
inline std::string object_of(const void* fn) {
  Dl_info di;
  if (fn && dladdr(fn, &di) && di.dli_fname) return di.dli_fname;
  return std::string();
}

inline std::string join_paths(const std::vector<std::string>& v) {
  std::string s;
  for (size_t i = 0; i < v.size(); i++) { if (i) s += " + "; s += v[i]; }
  return s;
}

// "" when the process holds at most one HIP runtime; otherwise what collides and what to do about it.  (Two libhsa-runtime64 objects
// under ONE HIP runtime are legitimate: rocprofv3's tool library links /opt/rocm's copy for its types and tables while the HIP runtime
// that registers with it initialises its own — every profile of this repository was taken that way.  It is a second HIP runtime that
// brings up a second HSA instance of its own.)
inline std::string runtime_conflict() {
  const RuntimeObjects o = mapped_runtime_objects();
  if (o.hip.size() <= 1) return std::string();
  std::string s = "two GPU runtimes are mapped into this process (HIP: " + join_paths(o.hip) + "; HSA: " + join_paths(o.hsa) +
                  "): only the one that initialises first gets the GPU.  PyTorch's wheel bundles its own copies under torch/lib; "
                  "load it BEFORE libptamd.so (the library then binds to torch's copy by soname), or load libptamd.so through "
                  "platinum_amd.abi.load_library(), which does that for you";
  return s;
}

}  // namespace pt
