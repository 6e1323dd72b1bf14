"""bench.py's image_to_objects leg on the 3 240-keypoint image, alone (for rocprofv3).  usage: image_leg_textured.py [frames=1024]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.image_to_objects_leg(argparse.Namespace(models=20), frames=int(sys.argv[1]) if len(sys.argv) > 1 else 1024, image="textured")
print(json.dumps({k: r[k] for k in ("value", "keypoints_per_image", "sift_alone_ms")}))
