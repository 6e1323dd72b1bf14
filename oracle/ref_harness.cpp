// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI wrapper over the parts of the reference hot path that compile
// from the reference's own sources without any external library:
//
//   * ANN 1.1.1 (float build) kd-tree 2-NN search -- the arithmetic under
//     MATCH_ANN_CPU (moped2/libmoped/src/match/MATCH_ANN_CPU.hpp:83-107,150-162;
//     libs.tgz -> ann_1.1.1/kd_search.cpp:89-119,172-210)
//   * levmar 2.4 slevmar_dif -- the optimiser under
//     POSE_RANSAC_LM_DIFF_REPROJECTION_CPU::optimizeCamera
//     (moped2/libmoped/src/pose/POSE_RANSAC_LM_DIFF_REPROJECTION_CPU.hpp:140-164;
//     libs.tgz -> levmar-2.4/lm_core.c:427-825)
//   * include/moped.hpp: Pt<N>, Pose, TransformMatrix, project()
//     (moped2/libmoped/include/moped.hpp:84-203,330-354)
//   * include/sXML.hpp + the stream operators of moped.hpp -- model file parsing under
//     Moped::addModel(sXML&) (moped2/libmoped/src/moped.cpp:101-137)
//   * libsiftfast 1.1 GetKeypoints -- only to produce real SIFT descriptors
//     for fixtures, called the way FEAT_SIFT_CPU does
//     (moped2/libmoped/src/feat/FEAT_SIFT_CPU.hpp:78-112).
//
// It is built by oracle/build_ref.sh from the sources where they lie under
// /root/reference (the vendored tarball is unpacked into a temp dir outside
// the repo) and the only output is oracle/_ref/libmoped_ref.so.
//
// NOT built from the reference: the STEP classes themselves
// (MATCH_ANN_CPU / CLUSTER_MEAN_SHIFT_CPU / POSE_RANSAC_* / FILTER_*). They
// need src/util.hpp, which includes OpenCV headers this image does not have
// (util.hpp:51-52); no stand-in headers are written, so those classes are
// treated as unbuildable here (see DESIGN.md "Oracle").  The small amount of
// glue this file adds around the libraries (the residual callback, the
// search loop) is our own restatement and says so where it appears.
//
// Build flags (see build_ref.sh): -std=gnu++98 (moped.hpp pulls std and
// tr1 into one namespace) and -fno-delete-null-pointer-checks (project()
// tests the address of a reference against NULL, moped.hpp:245,338).

#include <moped.hpp>
#include <ANN.h>
#include <lm.h>
#include <siftfast.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

using namespace MopedNS;

extern int DoubleImSize;  // libsiftfast.cpp:107 (FEAT_SIFT_CPU.hpp:50)

namespace {

struct AnnIndex {
  ANNpointArray pts;
  ANNkd_tree* tree;
  int n, dim;
};

// One correspondence as POSE_RANSAC_LM_DIFF_REPROJECTION_CPU::LmData holds it
// (…REPROJECTION_CPU.hpp:66-72): image (intrinsics + camera TM), 2-D, 3-D.
struct Corr {
  Image* image;
  Pt<2> coord2D;
  Pt<3> coord3D;
};

// Restatement of lmFuncQuat (…REPROJECTION_CPU.hpp:100-138) on top of the
// reference's own Pose / TransformMatrix arithmetic: squared pixel residuals,
// (-z+10) for points behind the camera.
void residual_cb(float* p, float* hx, int /*m*/, int n, void* adata) {
  std::vector<Corr>& c = *(std::vector<Corr>*)adata;
  Pose pose;
  pose.rotation.init(p);
  pose.rotation.norm();
  pose.translation.init(p + 4);
  TransformMatrix tm;
  tm.init(pose);
  for (int i = 0; i < n / 2; i++) {
    Pt<3> x;
    tm.transform(x, c[i].coord3D);
    c[i].image->TM.inverseTransform(x, x);
    const Pt<4>& K = c[i].image->intrinsicLinearCalibration;
    float u = x[0] / x[2] * K[0] + K[2];
    float v = x[1] / x[2] * K[1] + K[3];
    if (x[2] < 0) {
      hx[2 * i] = -x[2] + 10;
      hx[2 * i + 1] = -x[2] + 10;
    } else {
      float du = u - c[i].coord2D[0];
      float dv = v - c[i].coord2D[1];
      hx[2 * i] = du * du;
      hx[2 * i + 1] = dv * dv;
    }
  }
}

// moped3d's LmData (…BACKPROJECTION_DEPTH_CPU.hpp:72-80) and the two depth residual
// callbacks, restated over the reference's Pt / TransformMatrix arithmetic
// (mode 1: …BACKPROJECTION_DEPTH_CPU.hpp:108-190; mode 2: …REPROJECTION_DEPTH_CPU.hpp:106-216).
struct CorrD {
  Image* image;
  Pt<2> coord2D;
  Pt<3> coord3D;
  Pt<3> world3D;
  float cauchyWeight;
};
struct DepthProblem {
  std::vector<CorrD> c;
  int mode;
  float alpha;
};

void residual_depth_cb(float* lmPose, float* errors, int /*m*/, int nErrors, void* adata) {
  DepthProblem& P = *(DepthProblem*)adata;
  Pose pose;
  pose.rotation.init(lmPose);
  pose.rotation.norm();
  pose.translation.init(lmPose + 4);
  TransformMatrix tm;
  tm.init(pose);
  const int per = P.mode == 1 ? 2 : 3;
  for (int i = 0; i < nErrors / per; i++) {
    CorrD& d = P.c[i];
    Pt<3> p3D;
    tm.transform(p3D, d.coord3D);
    d.image->TM.inverseTransform(p3D, p3D);
    Pt<4> K = d.image->intrinsicLinearCalibration;
    float wi = d.cauchyWeight;
    float weight3D = (1 - P.alpha) * wi;
    if (P.mode == 1) {
      if (p3D[2] < 0) {
        errors[2 * i] = -p3D[2] + 10;
        errors[2 * i + 1] = -p3D[2] + 10;
      } else {
        float vx = d.world3D[0], vy = d.world3D[1], vz = d.world3D[2];
        float norm = sqrt(vx * vx + vy * vy + vz * vz);
        float nx = vx / norm, ny = vy / norm, nz = vz / norm;
        float dotProd = nx * p3D[0] + ny * p3D[1] + nz * p3D[2];
        Pt<3> pHat;
        pHat.init(nx * dotProd, ny * dotProd, nz * dotProd);
        float dxy = p3D.euclDist(pHat);
        float dz = d.world3D.euclDist(pHat);
        errors[2 * i] = dxy * dxy;
        errors[2 * i + 1] = dz * dz;
      }
      float weight2D = 1 - weight3D;
      errors[2 * i] *= weight2D;
      errors[2 * i + 1] *= weight3D;
    } else {
      float u = p3D[0] / p3D[2] * K[0] + K[2];
      float v = p3D[1] / p3D[2] * K[1] + K[3];
      if (p3D[2] < 0) {
        errors[3 * i] = -p3D[2] + 10;
        errors[3 * i + 1] = -p3D[2] + 10;
        errors[3 * i + 2] = -p3D[2] + 10;
      } else {
        float dx = u - d.coord2D[0], dy = v - d.coord2D[1];
        errors[3 * i] = dx * dx;
        errors[3 * i + 1] = dy * dy;
      }
      float vecTP = p3D[0] * d.world3D[0] + p3D[1] * d.world3D[1] + p3D[2] * d.world3D[2];
      Pt<3> projWorld;
      projWorld.init(p3D[0] * vecTP, p3D[1] * vecTP, p3D[2] * vecTP);
      float depthError = projWorld.euclDist(p3D);
      errors[3 * i + 2] = depthError * depthError;
      errors[3 * i + 2] *= 50;
      errors[3 * i] *= (1 - weight3D);
      errors[3 * i + 1] *= (1 - weight3D);
      errors[3 * i + 2] *= weight3D;
    }
  }
}

void fill_image(Image& img, const float K[4], const float cam[7]) {
  img.width = 640;
  img.height = 480;
  img.intrinsicLinearCalibration.init(K[0], K[1], K[2], K[3]);
  img.intrinsicNonlinearCalibration.init(0.f, 0.f, 0.f, 0.f);
  img.cameraPose.rotation.init(cam[0], cam[1], cam[2], cam[3]);
  img.cameraPose.translation.init(cam[4], cam[5], cam[6]);
  img.TM.init(img.cameraPose);  // as MopedPimpl::processImages does (moped.cpp:168-169)
}

}  // namespace

extern "C" {

// ---- ANN -------------------------------------------------------------------

// db: n x dim row-major, used as given (caller normalises like Update() :94).
void* ref_ann_build(const float* db, int n, int dim) {
  AnnIndex* ix = new AnnIndex;
  ix->n = n;
  ix->dim = dim;
  ix->pts = annAllocPts(n, dim);
  for (int i = 0; i < n; i++) memcpy(ix->pts[i], db + (size_t)i * dim, dim * sizeof(float));
  ix->tree = new ANNkd_tree(ix->pts, n, dim);  // defaults: bucket 1, ANN_KD_SUGGEST
  return ix;
}

// 2-NN per query with error bound eps (MATCH_ANN_CPU's Quality); squared
// distances out (ANN never takes the root).  idx/dist: 2 per query.
void ref_ann_search2(void* h, const float* q, int nq, float eps, int* idx, float* dist) {
  AnnIndex* ix = (AnnIndex*)h;
  ANNpoint pt = annAllocPt(ix->dim);
  ANNidx nx[2];
  ANNdist ds[2];
  for (int i = 0; i < nq; i++) {
    memcpy(pt, q + (size_t)i * ix->dim, ix->dim * sizeof(float));
    ix->tree->annkSearch(pt, 2, nx, ds, eps);
    idx[2 * i] = nx[0];
    idx[2 * i + 1] = nx[1];
    dist[2 * i] = ds[0];
    dist[2 * i + 1] = ds[1];
  }
  annDeallocPt(pt);
}

void ref_ann_free(void* h) {
  AnnIndex* ix = (AnnIndex*)h;
  delete ix->tree;
  annDeallocPts(ix->pts);
  delete ix;
  annClose();
}

// ---- project() / levmar ------------------------------------------------------

// Reference project() (moped.hpp:330-354).  pose/cam = (qx,qy,qz,qw,tx,ty,tz).
void ref_project(const float pose7[7], const float* xyz, int n, const float K[4],
                 const float cam[7], float* uv) {
  Image img;
  fill_image(img, K, cam);
  Pose pose;
  pose.rotation.init(pose7[0], pose7[1], pose7[2], pose7[3]);
  pose.translation.init(pose7[4], pose7[5], pose7[6]);
  for (int i = 0; i < n; i++) {
    Pt<3> X;
    X.init(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    Pt<2> p = project(pose, X, img);
    uv[2 * i] = p[0];
    uv[2 * i + 1] = p[1];
  }
}

// The residual vector lmFuncQuat would produce for (pose7, correspondences).
void ref_residuals(const float pose7[7], const float* uv, const float* xyz, int n,
                   const float K[4], const float cam[7], float* hx) {
  Image img;
  fill_image(img, K, cam);
  std::vector<Corr> c(n);
  for (int i = 0; i < n; i++) {
    c[i].image = &img;
    c[i].coord2D.init(uv[2 * i], uv[2 * i + 1]);
    c[i].coord3D.init(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  }
  float p[7];
  memcpy(p, pose7, sizeof p);
  residual_cb(p, hx, 7, 2 * n, &c);
}

// optimizeCamera (…REPROJECTION_CPU.hpp:140-164): slevmar_dif on 7 params with
// zero targets and default options; pose7 updated in place with the
// quaternion re-normalised; returns slevmar_dif's return value (iterations,
// or -1 on LM_ERROR) and info[LM_INFO_SZ].
int ref_optimize_camera(float pose7[7], const float* uv, const float* xyz, int n,
                        const float K[4], const float cam[7], int itmax, float* info) {
  Image img;
  fill_image(img, K, cam);
  std::vector<Corr> c(n);
  for (int i = 0; i < n; i++) {
    c[i].image = &img;
    c[i].coord2D.init(uv[2 * i], uv[2 * i + 1]);
    c[i].coord3D.init(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  }
  std::vector<float> target(2 * n, 0.f);
  float linfo[LM_INFO_SZ];
  int ret = slevmar_dif(residual_cb, pose7, &target[0], 7, 2 * n, itmax, NULL, linfo, NULL, NULL,
                        (void*)&c);
  if (info) memcpy(info, linfo, sizeof linfo);
  if (ret < 0) return ret;
  Quat q;
  q.init(pose7[0], pose7[1], pose7[2], pose7[3]);
  q.norm();
  for (int i = 0; i < 4; i++) pose7[i] = q[i];
  return ret;
}

// ---- moped3d depth residuals ----------------------------------------------------
static void fill_depth(DepthProblem& P, Image& img, int mode, float alpha, const float* uv, const float* xyz,
                       const float* world, const float* wgt, int n) {
  P.mode = mode;
  P.alpha = alpha;
  P.c.resize(n);
  for (int i = 0; i < n; i++) {
    P.c[i].image = &img;
    P.c[i].coord2D.init(uv[2 * i], uv[2 * i + 1]);
    P.c[i].coord3D.init(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    P.c[i].world3D.init(world[3 * i], world[3 * i + 1], world[3 * i + 2]);
    P.c[i].cauchyWeight = wgt[i];
  }
}

void ref_residuals_depth(int mode, const float pose7[7], const float* uv, const float* xyz,
                         const float* world, const float* wgt, int n, const float K[4],
                         const float cam[7], float alpha, float* err) {
  Image img;
  fill_image(img, K, cam);
  DepthProblem P;
  fill_depth(P, img, mode, alpha, uv, xyz, world, wgt, n);
  float p[7];
  memcpy(p, pose7, sizeof p);
  residual_depth_cb(p, err, 7, (mode == 1 ? 2 : 3) * n, &P);
}

int ref_optimize_camera_depth(int mode, float pose7[7], const float* uv, const float* xyz,
                              const float* world, const float* wgt, int n, const float K[4],
                              const float cam[7], float alpha, int itmax, float* info) {
  Image img;
  fill_image(img, K, cam);
  DepthProblem P;
  fill_depth(P, img, mode, alpha, uv, xyz, world, wgt, n);
  const int ne = (mode == 1 ? 2 : 3) * n;
  std::vector<float> target(ne, 0.f);
  float linfo[LM_INFO_SZ];
  int ret = slevmar_dif(residual_depth_cb, pose7, &target[0], 7, ne, itmax, NULL, linfo, NULL, NULL, (void*)&P);
  if (info) memcpy(info, linfo, sizeof linfo);
  if (ret < 0) return ret;
  Quat q;
  q.init(pose7[0], pose7[1], pose7[2], pose7[3]);
  q.norm();
  for (int i = 0; i < 4; i++) pose7[i] = q[i];
  return ret;
}

// ---- libsiftfast -------------------------------------------------------------

// gray: h x w bytes.  Keypoints in list order as FEAT_SIFT_CPU emits them:
// xy = (col,row), 128 floats each.  Returns the keypoint count (may exceed
// max_kp; only max_kp are written).
int ref_sift2(const unsigned char* gray, int w, int h, float* xy, float* scale_ori, float* desc, int max_kp);
int ref_sift(const unsigned char* gray, int w, int h, float* xy, float* desc, int max_kp) {
  return ref_sift2(gray, w, h, xy, NULL, desc, max_kp);
}
// the same, also returning (scale, orientation) of every keypoint
int ref_sift2(const unsigned char* gray, int w, int h, float* xy, float* scale_ori, float* desc, int max_kp) {
  DoubleImSize = 1;  // ScaleOrigin "-1" (config.hpp:69, FEAT_SIFT_CPU.hpp:72-73)
  SFImage image = CreateImage(h, w);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      image->pixels[y * image->stride + x] = ((float)gray[w * y + x]) * 1. / 255.;
  Keypoint keypts = GetKeypoints(image);
  int n = 0;
  for (Keypoint k = keypts; k; k = k->next, n++) {
    if (n >= max_kp) continue;
    xy[2 * n] = k->col;
    xy[2 * n + 1] = k->row;
    if (scale_ori) {
      scale_ori[2 * n] = k->scale;
      scale_ori[2 * n + 1] = k->ori;
    }
    memcpy(desc + (size_t)n * 128, k->descrip, 128 * sizeof(float));
  }
  FreeKeypoints(keypts);
  DestroyAllImages();
  return n;
}

// ---- model files --------------------------------------------------------------
// The reference's own XML tokenizer (include/sXML.hpp:53-118, pulled in by moped.hpp:52)
// and number parsing (istream >> Pt<3>, moped.hpp:125; istream >> Float), walked the way
// Moped::addModel(sXML&) walks them (src/moped.cpp:101-137).  addModel itself is a member
// of MopedPimpl in moped.cpp, which includes config.hpp -> OpenCV (unbuildable here), so
// the ~15-line walk below is OUR restatement of it; everything it calls is the reference's.
// Writes the points filed under `desc_type`: xyz[cap][3], desc[cap][dim] (first `dim`
// values of each descriptor), name, bbox = min xyz, max xyz.  Returns the number of such
// points (may exceed cap), -1 if the file does not parse, -2 if there is no <Points>.
int ref_model_xml(const char* path, const char* desc_type, float* xyz, float* desc, int cap, int dim,
                  char* name, int name_cap, float* bbox6, int* n_bad_len) {
  sXML sxml;
  std::string fn(path);
  if (!sxml.fromFile(fn)) return -1;
  std::string nm = sxml["name"];
  snprintf(name, name_cap, "%s", nm.c_str());
  Pt<3> lo, hi;
  lo.init(10E10, 10E10, 10E10);
  hi.init(-10E10, -10E10, -10E10);
  sXML* points = NULL;
  for (size_t i = 0; i < sxml.children.size(); i++)
    if (sxml.children[i].name == "Points") points = &sxml.children[i];
  if (points == NULL) return -2;
  int n = 0, bad = 0;
  for (size_t i = 0; i < points->children.size(); i++) {
    sXML& pt = points->children[i];
    Pt<3> c;
    c.init(0, 0, 0);
    std::istringstream iss(pt["p3d"]);
    iss >> c;
    lo.min(c);
    hi.max(c);
    std::vector<float> d;
    std::istringstream jss(pt["desc"]);
    Float f;
    while (jss >> f) d.push_back(f);
    if (pt["desc_type"] != desc_type) continue;
    if ((int)d.size() != dim) bad++;
    if (n < cap) {
      for (int k = 0; k < 3; k++) xyz[3 * n + k] = c[k];
      for (int k = 0; k < dim; k++) desc[(size_t)n * dim + k] = k < (int)d.size() ? d[k] : 0.f;
    }
    n++;
  }
  for (int k = 0; k < 3; k++) {
    bbox6[k] = lo[k];
    bbox6[3 + k] = hi[k];
  }
  if (n_bad_len) *n_bad_len = bad;
  return n;
}

}  // extern "C"
