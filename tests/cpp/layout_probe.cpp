// One node, two translation units: this file is compiled TWICE -- by hipcc with <thrust/...> first (the form whose getModel()
// returns device views) and by plain g++ without thrust (the host-copy form) -- and each build prints what it sees of
// include/ssf.hpp's data layout.  tests/test_cpp_wrapper.py asserts the two agree: sizeof(SupersurfelFusion), sizeof(Supersurfels),
// sizeof(Transform3) do not depend on the include order (a .hip file and a main.cpp may share one object), while the class's
// inline-namespace tag differs, so that passing the object between the two forms is a LINK error rather than a wrong getModel().
#ifdef LAYOUT_WITH_THRUST
#include <thrust/host_vector.h>
#endif
#include <cstdio>
#include <cstring>
#include <typeinfo>
#include "ssf.hpp"

int main() {
    using namespace supersurfel_fusion;
    const char* mangled = typeid(SupersurfelFusion).name();
#ifdef SSF_THRUST_VIEW
    const int form = 1;
    if (!std::strstr(mangled, "thrust_view")) return 5;
#else
    const int form = 0;
    if (!std::strstr(mangled, "host_copy")) return 5;
#endif
    std::printf("form=%d fusion=%zu views=%zu array=%zu transform=%zu float3=%zu\n", form, sizeof(SupersurfelFusion), sizeof(Supersurfels),
                sizeof(DeviceArray<Mat33>), sizeof(Transform3), sizeof(float3));
    return 0;
}
