// The analytic test scene of the C++ examples: the inside of a box room
// (V-shaped back wall) with a sphere in it, rendered on the host for a camera
// that looks down +z and is only translated. Poses are known in closed form,
// so the examples can check their own trajectories.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace analytic_room {

struct Camera {
    double fx, fy, cx, cy;
    int width, height;
};

// Distance along the viewing ray (direction (x, y, 1) in the camera frame, the
// camera looks down +z and is only translated) to the first surface; returns
// the z-depth in metres and the hit point in the world.
inline double CastRay(const double eye[3], double dx, double dy, double hit[3]) {
    const double dir[3] = {dx, dy, 1.0};
    // small enough that the side walls, floor and ceiling are inside the field
    // of view: a lone front wall + sphere would leave one rotation free
    const double lo[3] = {-1.2, -0.9, -1.0}, hi[3] = {1.3, 0.8, 2.2};
    double t = 1e30;
    for (int a = 0; a < 3; ++a) {
        if (dir[a] > 1e-12) t = std::fmin(t, (hi[a] - eye[a]) / dir[a]);
        if (dir[a] < -1e-12) t = std::fmin(t, (lo[a] - eye[a]) / dir[a]);
    }
    // the back of the room is a shallow V (two walls meeting at x = 0): their
    // normals carry an x component, which is what lets point-to-plane
    // tracking see the sideways motion
    for (int side = -1; side <= 1; side += 2) {
        const double n[3] = {0.6 * side, 0.0, 1.0}, d = 2.0;
        const double denom = n[0] * dir[0] + n[2] * dir[2];
        const double num = d - (n[0] * eye[0] + n[2] * eye[2]);
        if (denom > 1e-12) t = std::fmin(t, num / denom);
    }
    // sphere
    const double c[3] = {0.3, 0.2, 1.5}, r = 0.35;
    double oc[3] = {eye[0] - c[0], eye[1] - c[1], eye[2] - c[2]};
    const double A = dir[0] * dir[0] + dir[1] * dir[1] + 1.0;
    const double B = 2 * (oc[0] * dir[0] + oc[1] * dir[1] + oc[2] * dir[2]);
    const double Cq = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - r * r;
    const double disc = B * B - 4 * A * Cq;
    if (disc > 0) {
        const double ts = (-B - std::sqrt(disc)) / (2 * A);
        if (ts > 0 && ts < t) t = ts;
    }
    for (int a = 0; a < 3; ++a) hit[a] = eye[a] + t * dir[a];
    return t;  // dir.z == 1: the ray parameter is the z-depth
}

inline void RenderFrame(const Camera& cam, const double eye[3],
                 std::vector<uint16_t>& depth, std::vector<uint8_t>& color) {
    depth.resize((size_t)cam.width * cam.height);
    color.resize((size_t)cam.width * cam.height * 3);
    for (int v = 0; v < cam.height; ++v)
        for (int u = 0; u < cam.width; ++u) {
            double hit[3];
            const double z = CastRay(eye, (u - cam.cx) / cam.fx,
                                     (v - cam.cy) / cam.fy, hit);
            const double mm = z * 1000.0;
            const size_t i = (size_t)v * cam.width + u;
            depth[i] = mm < 65535.0 ? (uint16_t)(mm + 0.5) : 0;
            for (int a = 0; a < 3; ++a) {
                // smooth texture so that every pixel carries a gradient
                const double s = 0.5 + 0.5 * std::sin(5.0 * hit[a] + a);
                color[3 * i + a] = (uint8_t)(40 + 170 * s);
            }
        }
}

}  // namespace analytic_room
