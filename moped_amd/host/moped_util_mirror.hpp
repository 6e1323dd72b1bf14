// The part of the stand-in types that mirrors the reference's src/util.hpp (depthInformation, FrameData, MopedAlg and the
// pipeline container; util.hpp:68-201, moped3d: :73-107) -- and ONLY that part: it expects Pt / Pose / Image / Model /
// Object / Float / SP_* / toString of namespace MopedNS to be declared already, either by moped_types.hpp (stand-alone
// builds, the GPU box) or by the reference's REAL include/moped.hpp (`make check_ref`: the plugin headers are then
// compiled against the real Pt / Pose / Model / Image / Object layouts; util.hpp itself cannot be used because it
// includes OpenCV headers this image does not have, util.hpp:51-52).  Test scaffolding, not something a libmoped
// maintainer takes.
#pragma once
#include <list>
#include <map>
#include <string>
#include <vector>

namespace MopedNS {

#ifdef MOPED_AMD_WITH_DEPTH
// moped3d only (moped3d/libmoped/src/util.hpp:73-84): what DEPTHMAP_PROP_CPU attaches to a match
struct depthInformation {
  bool depthValid;
  Pt<3> coord3D;       // camera-frame xyz read from the depth map
  Float depth;
  Float fillDistance;  // distance to the pixel the depth was filled in from (-1: unknown)
};
#endif

struct FrameData {
  struct DetectedFeature {
    int imageIdx;
    Pt<2> coord2D;
    vector<float> descriptor;
  };
  struct Match {
    int imageIdx;
    Pt<2> coord2D;
    Pt<3> coord3D;
#ifdef MOPED_AMD_WITH_DEPTH
    depthInformation depthData;  // moped3d/libmoped/src/util.hpp:107
#endif
  };
  typedef list<int> Cluster;
  vector<SP_Image> images;
  map<string, vector<DetectedFeature> > detectedFeatures;
  vector<vector<Match> > matches;
  vector<vector<Cluster> > clusters;
  list<SP_Object>* objects;
  int correctMatches, incorrectMatches;
  vector<vector<Cluster> > oldClusters;
  list<SP_Object> oldObjects;
  map<string, Float> times;
};

class MopedAlg {
 public:
  vector<SP_Model>* models;
  bool capable;
  bool configUpdated;
  string _stepName;
  int _alg;
  MopedAlg() : models(0), capable(true), configUpdated(true), _alg(0) {}
  virtual ~MopedAlg() {}
  bool isCapable() const { return capable; }
  void setStepNameAndAlg(string& stepName, int alg) { _stepName = stepName; _alg = alg; }
  virtual void modelsUpdated(vector<SP_Model>& m) { models = &m; configUpdated = true; }
  virtual void getConfig(map<string, string>&) const {}
  virtual void setConfig(map<string, string>&) {}
  virtual void process(FrameData& frameData) = 0;
};

struct MopedStep : public vector<shared_ptr<MopedAlg> > {
  // the reference's fall-back hook: the first algorithm of the step that is capable
  MopedAlg* getAlg() {
    for (iterator it = begin(); it != end(); ++it)
      if ((*it)->isCapable()) return it->get();
    return 0;
  }
};

// Steps in first-seen order of their names; several algorithms may share a step name.
struct MopedPipeline : public vector<MopedStep> {
  vector<string> stepNames;   // stepNames[i] names (*this)[i]

  int stepIndex(const string& name) {
    for (size_t i = 0; i < stepNames.size(); ++i)
      if (stepNames[i] == name) return (int)i;
    stepNames.push_back(name);
    push_back(MopedStep());
    return (int)stepNames.size() - 1;
  }
  void addAlg(string stepName, MopedAlg* alg) {
    MopedStep& step = (*this)[stepIndex(stepName)];
    alg->setStepNameAndAlg(stepName, (int)step.size());
    step.push_back(shared_ptr<MopedAlg>(alg));   // the pipeline owns its algorithms
  }
  list<MopedAlg*> getAlgs(bool onlyActive = false) {
    list<MopedAlg*> out;
    for (iterator st = begin(); st != end(); ++st) {
      if (onlyActive) {
        if (MopedAlg* a = st->getAlg()) out.push_back(a);
        continue;
      }
      for (MopedStep::iterator a = st->begin(); a != st->end(); ++a) out.push_back(a->get());
    }
    return out;
  }
};

}  // namespace MopedNS
