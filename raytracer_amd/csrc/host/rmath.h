// Host-side f64 math for the scene side of the renderer: points/vectors,
// 4x4 row-major matrices, the approximate-equality predicate and the RNG.
//
// Every expression keeps the association order of the reference so that the
// matrices handed to the device are bit-identical to the reference's:
//   reference src/math/vector.rs:39-71, src/math/point.rs:24-70,
//   src/math/transformation/matrix.rs:24-45,99-123,133-245,255-318,
//   src/math/transformation/mod.rs:37-86, src/math/float.rs:10-24.
// Build with -ffp-contract=off (Rust never fuses a*b+c).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>

namespace rt {

constexpr double EPS = 2.220446049250313e-16;  // f64::EPSILON

// float-cmp 0.9.0 `ApproxEq for f64` with F64Margin::default()
// (epsilon = f64::EPSILON, ulps = 4); reference src/math/float.rs:10-24.
inline bool approx_eq(double a, double b) {
    if (a == b) return true;
    if (std::fabs(a - b) <= EPS) return true;
    int64_t ai, bi;
    std::memcpy(&ai, &a, 8);
    std::memcpy(&bi, &b, 8);
    int64_t d = (int64_t)((uint64_t)ai - (uint64_t)bi);  // wrapping_sub
    if (d == INT64_MIN) return false;                     // saturating abs
    if (d < 0) d = -d;
    return d <= 4;
}

struct V3 {
    double x = 0, y = 0, z = 0;
};

inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double magnitude(V3 a) { return std::sqrt(dot(a, a)); }
// reference src/math/vector.rs:44-52
inline V3 normalise(V3 a) {
    double m = magnitude(a);
    if (approx_eq(m, 0.0)) return {0, 0, 0};
    return a / m;
}
// reference src/math/vector.rs:69-71: v - (n * 2.0) * dot(v, n)
inline V3 reflect(V3 v, V3 n) { return v - (n * 2.0) * dot(v, n); }
inline bool approx_eq(V3 a, V3 b) {
    return approx_eq(a.x, b.x) && approx_eq(a.y, b.y) && approx_eq(a.z, b.z);
}

struct Mat4 {
    double m[4][4];

    static Mat4 zero() {
        Mat4 r;
        std::memset(r.m, 0, sizeof r.m);
        return r;
    }
    static Mat4 identity() {
        Mat4 r = zero();
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
        return r;
    }
    static Mat4 translate(double x, double y, double z) {
        Mat4 r = identity();
        r.m[0][3] = x; r.m[1][3] = y; r.m[2][3] = z;
        return r;
    }
    static Mat4 scale(double x, double y, double z) {
        Mat4 r = identity();
        r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z;
        return r;
    }
    static Mat4 rotate_x(double a) {
        double s = std::sin(a), c = std::cos(a);
        Mat4 r = identity();
        r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c;
        return r;
    }
    static Mat4 rotate_y(double a) {
        double s = std::sin(a), c = std::cos(a);
        Mat4 r = identity();
        r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c;
        return r;
    }
    static Mat4 rotate_z(double a) {
        double s = std::sin(a), c = std::cos(a);
        Mat4 r = identity();
        r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c;
        return r;
    }
    static Mat4 shear(double xy, double xz, double yx, double yz, double zx, double zy) {
        Mat4 r = identity();
        r.m[0][1] = xy; r.m[0][2] = xz; r.m[1][0] = yx; r.m[1][2] = yz;
        r.m[2][0] = zx; r.m[2][1] = zy;
        return r;
    }
    Mat4 transpose() const {
        Mat4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[j][i] = m[i][j];
        return r;
    }
};

// reference matrix.rs:255-271: accumulation starts from 0.0 and adds the four
// products in index order.
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 r = Mat4::zero();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) r.m[i][j] += a.m[i][k] * b.m[k][j];
    return r;
}

// reference matrix.rs:279-298 (row 3 is never used; the w term is m[r][3]).
inline V3 mul_point(const Mat4& a, V3 p) {
    return {a.m[0][0] * p.x + a.m[0][1] * p.y + a.m[0][2] * p.z + a.m[0][3],
            a.m[1][0] * p.x + a.m[1][1] * p.y + a.m[1][2] * p.z + a.m[1][3],
            a.m[2][0] * p.x + a.m[2][1] * p.y + a.m[2][2] * p.z + a.m[2][3]};
}
// reference matrix.rs:308-318
inline V3 mul_vector(const Mat4& a, V3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
            a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}

// Cofactor expansion along row 0, exactly as reference matrix.rs:64-126,248-253.
inline double det2(const double a[2][2]) { return a[0][0] * a[1][1] - a[0][1] * a[1][0]; }
inline double det3(const double a[3][3]);
inline double cofactor3(const double a[3][3], int row, int col) {
    double s[2][2];
    int nr = 0;
    for (int r = 0; r < 3; r++) {
        if (r == row) continue;
        int nc = 0;
        for (int c = 0; c < 3; c++) {
            if (c == col) continue;
            s[nr][nc++] = a[r][c];
        }
        nr++;
    }
    double minor = det2(s);
    return ((row + col) % 2 != 0) ? minor * -1.0 : minor;
}
inline double det3(const double a[3][3]) {
    double det = 0.0;
    for (int c = 0; c < 3; c++) det += a[0][c] * cofactor3(a, 0, c);
    return det;
}
inline double cofactor4(const Mat4& a, int row, int col) {
    double s[3][3];
    int nr = 0;
    for (int r = 0; r < 4; r++) {
        if (r == row) continue;
        int nc = 0;
        for (int c = 0; c < 4; c++) {
            if (c == col) continue;
            s[nr][nc++] = a.m[r][c];
        }
        nr++;
    }
    double minor = det3(s);
    return ((row + col) % 2 != 0) ? minor * -1.0 : minor;
}
inline double det4(const Mat4& a) {
    double det = 0.0;
    for (int c = 0; c < 4; c++) det += a.m[0][c] * cofactor4(a, 0, c);
    return det;
}
// reference matrix.rs:227-245; a singular matrix is a panic there.
inline Mat4 invert(const Mat4& a) {
    double det = det4(a);
    if (approx_eq(det, 0.0))
        throw std::runtime_error("Tried to invert a Matrix that cannot be inverted");
    Mat4 r = Mat4::zero();
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) r.m[col][row] = cofactor4(a, row, col) / det;
    return r;
}
inline bool approx_eq(const Mat4& a, const Mat4& b) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (!approx_eq(a.m[i][j], b.m[i][j])) return false;
    return true;
}

// reference matrix.rs:206-220
inline Mat4 view_transformation(V3 from, V3 to, V3 up) {
    V3 forward = normalise(to - from);
    V3 upn = normalise(up);
    V3 left = cross(forward, upn);
    V3 true_up = cross(left, forward);
    Mat4 o = Mat4::identity();
    o.m[0][0] = left.x; o.m[0][1] = left.y; o.m[0][2] = left.z;
    o.m[1][0] = true_up.x; o.m[1][1] = true_up.y; o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x; o.m[2][1] = -forward.y; o.m[2][2] = -forward.z;
    return o * Mat4::translate(-from.x, -from.y, -from.z);
}

// Xoshiro256PlusPlus::seed_from_u64 + gen::<u64>() + gen_range(lo..=hi) of
// rand_xoshiro 0.6.0 / rand 0.8.5 (third-party crates pinned in the
// reference's Cargo.lock:660-691; call sites src/camera.rs:144,159,
// src/light/area.rs:50-51, src/scene/mod.rs:169-206). Pinned by the known
// answers in reference src/light/area.rs:136-160.
struct Xoshiro256pp {
    uint64_t s[4];

    static Xoshiro256pp seed_from_u64(uint64_t state) {
        Xoshiro256pp r;
        for (int i = 0; i < 4; i++) {  // SplitMix64
            state += 0x9e3779b97f4a7c15ULL;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            r.s[i] = z ^ (z >> 31);
        }
        return r;
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next_u64() {
        uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    // UniformFloat<f64>::new_inclusive(lo, hi).sample(rng)
    double gen_range_inclusive(double lo, double hi) {
        const double max_rand = 1.0 - 0x1p-52;  // (u64::MAX >> 12) as [1,2) float, minus 1
        double scale = (hi - lo) / max_rand;
        while (scale * max_rand + lo > hi) {
            uint64_t b;
            std::memcpy(&b, &scale, 8);
            b -= 1;
            std::memcpy(&scale, &b, 8);
        }
        uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ULL;
        double v12;
        std::memcpy(&v12, &bits, 8);
        double v01 = v12 - 1.0;
        return v01 * scale + lo;
    }
};

}  // namespace rt
