// include/ssf.hpp in a translation unit that already has HIP's vector types (a .hip / hipcc-compiled node, or any file that
// includes <hip/hip_runtime.h> first): the header must NOT declare its own float3 -- Transform3 / Mat33 are then built on HIP's
// float3 (same three floats x, y, z), make_float3 keeps working, and the reference nodes' pose lines still compile.
// Compile-only (tests/test_cpp_wrapper.py): g++ -c with /opt/rocm/include on the path.
#include <hip/hip_vector_types.h>
#include "ssf.hpp"
#include "tf_double.hpp"
using namespace supersurfel_fusion;
#ifdef SSF_HAVE_FLOAT3
#error "ssf.hpp declared its own float3 although HIP's vector types were already there"
#endif
static_assert(sizeof(float3) == 12 && sizeof(Mat33) == 36 && sizeof(Transform3) == 48, "matrix_types.h layout on HIP's float3");
double pose_sum(SupersurfelFusion& ssf) {
    Transform3 pose = ssf.getPose();
    tf::Transform opt_to_map(tf::Matrix3x3(pose.R.rows[0].x, pose.R.rows[0].y, pose.R.rows[0].z,
                                           pose.R.rows[1].x, pose.R.rows[1].y, pose.R.rows[1].z,
                                           pose.R.rows[2].x, pose.R.rows[2].y, pose.R.rows[2].z),
                             tf::Vector3(pose.t.x, pose.t.y, pose.t.z));
    const float3 t = make_float3(pose.t.x, pose.t.y, pose.t.z);           // HIP's own constructor function on the member's type
    return opt_to_map.getOrigin().x() + (double)t.y;
}
