// One mh_ctx shared by the HIP steps of a pipeline (the steps run strictly one
// after another on the caller's thread, src/moped.cpp:184-191).  C++98-clean and
// free of HIP headers so it compiles inside libmoped with its own flags
// (-std=gnu++98, libmoped/Makefile:44-47); the GPU lives behind the C ABI.
#pragma once
#include <moped_hip.h>

#include <iostream>

namespace MopedNS {

class HipSession {
 public:
  // NULL when no gfx950 device / library problem: the step then sets capable=false
  // and MopedStep::getAlg() falls through to the next algorithm of the same step
  // (src/util.hpp:151-159).
  static mh_ctx* get(int device = 0) {
    static HipSession s(device);
    return s.ctx_;
  }
  static void warn(const char* where) {
    mh_ctx* c = get();
    std::clog << "[moped_hip] " << where << ": " << (c ? mh_last_error(c) : "no context") << std::endl;
  }

 private:
  explicit HipSession(int device) : ctx_(0) {
    if (mh_create(device, &ctx_) != MH_OK) ctx_ = 0;
  }
  ~HipSession() {
    if (ctx_) mh_destroy(ctx_);
  }
  mh_ctx* ctx_;
};

// Same key layout as GET_CONFIG: "<STEP>:<algIdx>:<HeaderBasename>/<var>" (src/util.hpp:62)
template <typename T>
inline void hipGetConfig(std::map<std::string, std::string>& config, const std::string& step, int alg,
                         const char* header, const char* var, const T& value) {
  config[step + ":" + toString(alg) + ":" + header + "/" + var] = toString(value);
}

}  // namespace MopedNS
