// FILTER_PROJECTION_HIP -- drop-in for FILTER_PROJECTION_CPU
// (src/filter/FILTER_PROJECTION_CPU.hpp), same constructor arguments:
//     pipeline.addAlg( "FILTER",  new FILTER_PROJECTION_HIP( 5, 4096., 2 ) );
//     pipeline.addAlg( "FILTER2", new FILTER_PROJECTION_HIP( 7, 4096., 3 ) );
// Scores every object, gives each keypoint to the best-scoring object, erases
// objects with too few points / too low a score and rewrites frameData.clusters
// (:80-162).  Every match is projected through the image it came from and the ownership map is keyed by
// (coord2D, image) (:100-141), whatever else FrameData::images holds (a moped3d frame lists its depth and
// distance maps there too; moped3d wires this same step, moped3d/libmoped/src/config.hpp:48,50).
#pragma once
#include "hip_session.hpp"

namespace MopedNS {

class FILTER_PROJECTION_HIP : public MopedAlg {
  int MinPoints;
  Float FeatureDistance;
  Float MinScore;

 public:
  FILTER_PROJECTION_HIP(int MinPoints, Float FeatureDistance)
      : MinPoints(MinPoints), FeatureDistance(FeatureDistance), MinScore(0) {
    capable = HipSession::get() != 0;
  }
  FILTER_PROJECTION_HIP(int MinPoints, Float FeatureDistance, Float MinScore)
      : MinPoints(MinPoints), FeatureDistance(FeatureDistance), MinScore(MinScore) {
    capable = HipSession::get() != 0;
  }

  void getConfig(map<string, string>& config) const {
    hipGetConfig(config, _stepName, _alg, "FILTER_PROJECTION_HIP", "MinPoints", MinPoints);
    hipGetConfig(config, _stepName, _alg, "FILTER_PROJECTION_HIP", "FeatureDistance", FeatureDistance);
    hipGetConfig(config, _stepName, _alg, "FILTER_PROJECTION_HIP", "MinScore", MinScore);
  }
  void setConfig(map<string, string>&) {}

  // FILTER / FILTER2 on the objects POSE_RANSAC_P3P_HIP left on the device (HipHandover); false = not taken
  bool processResident(FrameData& frameData) {
    HipHandover& ho = HipHandover::get();
    const int which = ho.at(2, frameData) ? 1 : ho.at(4, frameData) ? 2 : 0;
    if (!which) return false;
    if (ho.matchesTag != HipHandover::tagMatches(frameData) || ho.objectsTag != HipHandover::tagObjects(frameData)) return false;
    const int nm = (int)models->size();
    // the device's objects are the list's, in list order
    vector<list<SP_Object>::iterator> its;
    vector<int> objModel;
    for (list<SP_Object>::iterator it = frameData.objects->begin(); it != frameData.objects->end(); ++it) {
      int m = 0;
      while (m < nm && (*models)[m].get() != (*it)->model.get()) ++m;
      if (m == nm) return false;
      its.push_back(it);
      objModel.push_back(m);
    }
    const int nobj = (int)its.size();
    size_t total = 0;
    for (int m = 0; m < nm; ++m) total += frameData.matches[m].size();
    vector<float> score(nobj + 1);
    vector<uint8_t> keep(nobj + 1);
    vector<int32_t> order(nobj + 1), members(total + 1), cloff(nobj + 1);
    int32_t kept = 0;
    if (mh_step_filter(HipSession::get(), which, MinPoints, FeatureDistance, MinScore, nobj, &score[0], &keep[0], &order[0],
                       &members[0], &cloff[0], (int)total + 1, &kept) != MH_OK) {
      HipSession::warn("mh_step_filter");
      return false;
    }
    frameData.clusters.clear();
    frameData.clusters.resize(nm);
    for (int o = 0; o < nobj; ++o) (*its[o])->score = score[o];
    for (int k = 0; k < kept; ++k) {
      const int o = order[k];
      vector<FrameData::Cluster>& dst = frameData.clusters[objModel[o]];
      dst.resize(dst.size() + 1);
      for (int j = cloff[k]; j < cloff[k + 1]; ++j) dst.back().push_back(members[j]);
    }
    for (int o = 0; o < nobj; ++o)
      if (!keep[o]) frameData.objects->erase(its[o]);
    ho.stage = which == 1 ? 3 : 5;
    ++ho.taken;
    ho.clustersTag = HipHandover::tagClusters(frameData);
    ho.objectsTag = HipHandover::tagObjects(frameData);
    return true;
  }

  void process(FrameData& frameData) {
    vector<vector<FrameData::Match> >& matches = frameData.matches;
    if (matches.size() < models->size()) return;  // the reference's sanity check (:85-87)
    if (processResident(frameData)) return;
    HipHandover::get().drop();
    const int nm = (int)models->size();
    const HipCameraTable table(frameData);
    if (!table.ok) return;
    vector<mh_corr> corr;
    vector<int32_t> imageOf;
    vector<int32_t> off(nm + 1, 0);
    for (int m = 0; m < nm; ++m) {
      for (size_t k = 0; k < matches[m].size(); ++k) {
        mh_corr c;
        c.u = matches[m][k].coord2D[0]; c.v = matches[m][k].coord2D[1];
        c.x = matches[m][k].coord3D[0]; c.y = matches[m][k].coord3D[1]; c.z = matches[m][k].coord3D[2];
        corr.push_back(c);
        imageOf.push_back(table.local[matches[m][k].imageIdx]);
      }
      off[m + 1] = (int32_t)corr.size();
    }
    // objects in (model, list) order -- the order the reference's double loop visits them (:94-96)
    vector<list<SP_Object>::iterator> its;
    vector<int32_t> objModel;
    vector<float> objPose;
    for (int m = 0; m < nm; ++m)
      for (list<SP_Object>::iterator it = frameData.objects->begin(); it != frameData.objects->end(); ++it)
        if ((*it)->model->name == (*models)[m]->name) {
          its.push_back(it);
          objModel.push_back(m);
          for (int i = 0; i < 4; ++i) objPose.push_back((*it)->pose.rotation[i]);
          for (int i = 0; i < 3; ++i) objPose.push_back((*it)->pose.translation[i]);
        }
    const int nobj = (int)its.size();
    frameData.clusters.clear();
    frameData.clusters.resize(nm);
    if (nobj == 0) return;
    if (corr.empty()) {  // no match anywhere: every object scores 0 and goes (:143-160 with empty newClusters)
      for (int o = 0; o < nobj; ++o) {
        (*its[o])->score = 0;
        if (MinPoints > 0 || MinScore > 0) frameData.objects->erase(its[o]);
        else frameData.clusters[objModel[o]].push_back(FrameData::Cluster());
      }
      return;
    }
    vector<float> score(nobj);
    vector<uint8_t> keep(nobj);
    vector<int32_t> order(nobj), members(corr.size() + 1), cloff(nobj + 1);
    int32_t kept = 0;
    if (mh_filter_images(HipSession::get(), &corr[0], &imageOf[0], &off[0], nm, &objModel[0], &objPose[0], nobj,
                         &table.cams[0], (int)table.cams.size(), MinPoints, FeatureDistance, MinScore, &score[0],
                         &keep[0], &order[0], &members[0], &cloff[0], &kept) != MH_OK) {
      HipSession::warn("mh_filter_images");
      return;
    }
    for (int o = 0; o < nobj; ++o) (*its[o])->score = score[o];
    for (int k = 0; k < kept; ++k) {
      const int o = order[k];
      FrameData::Cluster cl;
      for (int j = cloff[k]; j < cloff[k + 1]; ++j) cl.push_back(members[j]);
      frameData.clusters[objModel[o]].push_back(cl);
    }
    for (int o = 0; o < nobj; ++o)
      if (!keep[o]) frameData.objects->erase(its[o]);
  }
};

}  // namespace MopedNS
