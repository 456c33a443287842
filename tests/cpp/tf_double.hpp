// tf_double.hpp -- a test double of the three tf:: types the reference nodes build from getPose()
// (node/supersurfel_fusion_node.cpp:88-91, node/supersurfel_fusion_rgbd_benchmark_node.cpp:617-620): tf::Matrix3x3 from
// nine scalars (row-major), tf::Vector3 from three, tf::Transform from both.  tfScalar is double, as in ROS.
// Test infrastructure only: a node includes <tf/transform_broadcaster.h> instead.
#pragma once
namespace tf {
typedef double tfScalar;
class Vector3 {
public:
    Vector3() : v_{0, 0, 0} {}
    Vector3(const tfScalar& x, const tfScalar& y, const tfScalar& z) : v_{x, y, z} {}
    const tfScalar& x() const { return v_[0]; }
    const tfScalar& y() const { return v_[1]; }
    const tfScalar& z() const { return v_[2]; }
private:
    tfScalar v_[3];
};
class Matrix3x3 {
public:
    Matrix3x3(const tfScalar& xx, const tfScalar& xy, const tfScalar& xz, const tfScalar& yx, const tfScalar& yy, const tfScalar& yz,
              const tfScalar& zx, const tfScalar& zy, const tfScalar& zz) : r_{Vector3(xx, xy, xz), Vector3(yx, yy, yz), Vector3(zx, zy, zz)} {}
    const Vector3& getRow(int i) const { return r_[i]; }
private:
    Vector3 r_[3];
};
class Transform {
public:
    explicit Transform(const Matrix3x3& b, const Vector3& c = Vector3()) : basis_(b), origin_(c) {}
    const Matrix3x3& getBasis() const { return basis_; }
    const Vector3& getOrigin() const { return origin_; }
private:
    Matrix3x3 basis_;
    Vector3 origin_;
};
}  // namespace tf
