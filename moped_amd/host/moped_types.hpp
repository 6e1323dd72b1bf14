// Stand-in declaration of the libmoped types the STEP plugins are written against, for
// building OUTSIDE a libmoped tree (the GPU box has no reference).  Inside libmoped,
// include the real <moped.hpp> and <util.hpp> instead and do NOT include this file: it is
// test scaffolding, not something a libmoped maintainer takes.
// It mirrors an interface, so names, member order and meaning are the reference's by
// necessity (include/moped.hpp:84-290, src/util.hpp:68-201, read as text); only what the
// HIP steps and the test harness touch is declared, and the pipeline container below is
// our own implementation of the same contract (first-seen step order, first capable
// algorithm of a step; util.hpp:151-201).
#pragma once
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace MopedNS {

using std::list;
using std::map;
using std::string;
using std::vector;
using std::shared_ptr;

typedef float Float;

template <int N>
struct Pt {
  Float p[N];
  Float& operator[](int n) { return p[n]; }
  const Float& operator[](int n) const { return p[n]; }
  template <typename T> Pt<N>& init(T a, T b) { p[0] = a; p[1] = b; return *this; }
  template <typename T> Pt<N>& init(T a, T b, T c) { p[0] = a; p[1] = b; p[2] = c; return *this; }
  template <typename T> Pt<N>& init(T a, T b, T c, T d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; return *this; }
};
typedef Pt<4> Quat;

struct Pose {
  Quat rotation;      // (x, y, z, w)
  Pt<3> translation;
};

struct Model {
  struct IP {
    Pt<3> coord3D;
    vector<float> descriptor;
  };
  string name;
  map<string, vector<IP> > IPs;
  Pt<3> boundingBox[2];
};
typedef shared_ptr<Model> SP_Model;

#ifdef MOPED_AMD_WITH_DEPTH
// moped3d only (moped3d/libmoped/include/moped.hpp:230): what an Image holds
enum Image_Type { IMAGE_TYPE_GRAY_IMAGE, IMAGE_TYPE_RGB_IMAGE, IMAGE_TYPE_DEPTH_MAP, IMAGE_TYPE_PROB_MAP };
#endif

struct Image {
#ifdef MOPED_AMD_WITH_DEPTH
  Image_Type imageType;
  Image() : imageType(IMAGE_TYPE_GRAY_IMAGE) {}
  // depth map: 4 floats per pixel (x, y, z, norm), getDepth = z; probability / distance map: 1 float
  // per pixel (moped3d/libmoped/include/moped.hpp:261-284)
  Float getDepth(int x, int y) const {
    Float v;
    std::memcpy(&v, &data[((size_t)y * width + x) * 4 * sizeof(Float) + 2 * sizeof(Float)], sizeof v);
    return v;
  }
  Float getProb(int x, int y) const {
    Float v;
    std::memcpy(&v, &data[((size_t)y * width + x) * sizeof(Float)], sizeof v);
    return v;
  }
#endif
  vector<unsigned char> data;
  string name;
  int width, height;
  Pt<4> intrinsicLinearCalibration;     // fx, fy, cx, cy
  Pt<4> intrinsicNonlinearCalibration;
  Pose cameraPose;
};
typedef shared_ptr<Image> SP_Image;

struct Object {
  SP_Model model;
  Pose pose;
  Float score;
};
typedef shared_ptr<Object> SP_Object;

template <typename T>
inline string toString(const T& v) {
  std::ostringstream o;
  o << v;
  return o.str();
}

}  // namespace MopedNS

#include "moped_util_mirror.hpp"
