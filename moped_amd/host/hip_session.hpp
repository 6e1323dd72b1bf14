// One mh_ctx shared by the HIP steps of a pipeline (the steps run strictly one
// after another on the caller's thread, src/moped.cpp:184-191).  C++98-clean and
// free of HIP headers so it compiles inside libmoped with its own flags
// (-std=gnu++98, libmoped/Makefile:44-47); the GPU lives behind the C ABI.
#pragma once
#include <moped_hip.h>

#include <cstdlib>
#include <iostream>
#include <list>
#include <map>
#include <sstream>
#include <string>
#include <vector>

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

// The hand-over between consecutive HIP steps of one frame (mh_step_*, include/moped_hip.h): a HIP step that finds
// FrameData as the HIP step before it left it -- same frame object, and the lists it is about to read hash to what that
// step wrote -- runs on the device-resident copy instead of uploading them again.  Anything else (a CPU step in between
// that touched the lists, a frame with several cameras, lists that were not empty before MATCH, MH_STEP_HANDOVER=0)
// takes the upload path of the slot, which is always valid; so does a refusal by the library (the context's frame
// arrays were used by another call).  One instance per process, like the context.
struct HipHandover {
  int stage;                 // -1: nothing resident; 0 MATCH .. 5 FILTER2 = the last slot that ran on the resident frame
  const void* frame;         // the FrameData it belongs to
  unsigned long long matchesTag, clustersTag, objectsTag;
  unsigned long taken;       // steps that ran on the resident frame so far (tests, moped_hip_test)

  static HipHandover& get() {
    static HipHandover h;
    return h;
  }
  static bool enabled() {
    static const int on = (std::getenv("MH_STEP_HANDOVER") && std::getenv("MH_STEP_HANDOVER")[0] == '0') ? 0 : 1;
    return on != 0;
  }
  void drop() { stage = -1; frame = 0; }

  // FNV-1a over the bytes a step reads
  static unsigned long long mix(unsigned long long h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
  }
  static unsigned long long tagMatches(const FrameData& fd) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t m = 0; m < fd.matches.size(); ++m) {
      const size_t n = fd.matches[m].size();
      h = mix(h, &n, sizeof n);
      for (size_t k = 0; k < n; ++k) {
        const FrameData::Match& x = fd.matches[m][k];
        const float v[5] = {(float)x.coord2D[0], (float)x.coord2D[1], (float)x.coord3D[0], (float)x.coord3D[1], (float)x.coord3D[2]};
        h = mix(h, &x.imageIdx, sizeof x.imageIdx);
        h = mix(h, v, sizeof v);
      }
    }
    return h;
  }
  static unsigned long long tagClusters(const FrameData& fd) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t m = 0; m < fd.clusters.size(); ++m) {
      const size_t n = fd.clusters[m].size();
      h = mix(h, &n, sizeof n);
      for (size_t c = 0; c < n; ++c) {
        const size_t sz = fd.clusters[m][c].size();
        h = mix(h, &sz, sizeof sz);
        for (FrameData::Cluster::const_iterator it = fd.clusters[m][c].begin(); it != fd.clusters[m][c].end(); ++it) h = mix(h, &*it, sizeof(int));
      }
    }
    return h;
  }
  static unsigned long long tagObjects(const FrameData& fd) {
    unsigned long long h = 1469598103934665603ull;
    for (list<SP_Object>::const_iterator it = fd.objects->begin(); it != fd.objects->end(); ++it) {
      const void* model = (*it)->model.get();
      float v[7];
      for (int i = 0; i < 4; ++i) v[i] = (float)(*it)->pose.rotation[i];
      for (int i = 0; i < 3; ++i) v[4 + i] = (float)(*it)->pose.translation[i];
      h = mix(h, &model, sizeof model);
      h = mix(h, v, sizeof v);
    }
    return h;
  }
  bool at(int wanted, const FrameData& fd) const { return enabled() && stage == wanted && frame == (const void*)&fd; }

 private:
  HipHandover() : stage(-1), frame(0), matchesTag(0), clustersTag(0), objectsTag(0), taken(0) {}
};

// The cameras of a frame for the *_images entry points.  The reference projects every match through
// *frameData.images[match.imageIdx] whatever else the image list holds (FILTER_PROJECTION_CPU.hpp:100-104,
// POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:228-237) -- a moped3d frame carries its depth and distance maps as
// further Images that no match points to -- so the table holds the images the matches REFER to, renumbered in
// image order (which keeps CLUSTER's and FILTER's per-image ordering), and `local` maps imageIdx to that number.
struct HipCameraTable {
  std::vector<mh_cam> cams;
  std::vector<int> local;  // imageIdx -> index into cams, -1 = no match refers to it
  bool ok;

  explicit HipCameraTable(const FrameData& frameData) : local(frameData.images.size(), -1), ok(true) {
    for (size_t m = 0; m < frameData.matches.size() && ok; ++m)
      for (size_t k = 0; k < frameData.matches[m].size(); ++k) {
        const int i = frameData.matches[m][k].imageIdx;
        if (i < 0 || i >= (int)local.size()) { ok = false; break; }
        local[i] = 0;
      }
    for (size_t i = 0; i < local.size() && ok; ++i) {
      if (local[i] < 0) continue;
      local[i] = (int)cams.size();
      const Image& im = *frameData.images[i];
      mh_cam c;
      for (int j = 0; j < 4; ++j) c.K[j] = im.intrinsicLinearCalibration[j];
      for (int j = 0; j < 4; ++j) c.cam[j] = im.cameraPose.rotation[j];
      for (int j = 0; j < 3; ++j) c.cam[4 + j] = im.cameraPose.translation[j];
      cams.push_back(c);
    }
    if (ok && (int)cams.size() > MH_MAX_IMAGES) ok = false;
    if (!ok)
      std::clog << "[moped_hip] frame refers to an image outside FrameData::images or to more than " << MH_MAX_IMAGES
                << " images: step skipped" << std::endl;
  }
};

// Same key layout as GET_CONFIG: "<STEP>:<algIdx>:<HeaderBasename>/<var>" (src/util.hpp:62)
template <typename T>
inline void hipGetConfig(std::map<std::string, std::string>& config, const std::string& step, int alg,
                         const char* header, const char* var, const T& value) {
  config[step + ":" + toString(alg) + ":" + header + "/" + var] = toString(value);
}

// The counterpart for setConfig: takes `value` from the key hipGetConfig writes; true if it was there and parsed.
// (The reference's SET_CONFIG builds its key with another substring length than GET_CONFIG, src/util.hpp:62-63, so a
// value set through Moped::setConfig never reaches a CPU step -- SURVEY F5; the HIP steps honour the key they publish.)
template <typename T>
inline bool hipSetConfig(std::map<std::string, std::string>& config, const std::string& step, int alg, const char* header,
                         const char* var, T& value) {
  std::map<std::string, std::string>::iterator it = config.find(step + ":" + toString(alg) + ":" + header + "/" + var);
  if (it == config.end() || it->second.empty()) return false;
  std::istringstream in(it->second);
  T v;
  if (!(in >> v)) return false;
  value = v;
  return true;
}

}  // namespace MopedNS
