// TEST INFRASTRUCTURE (oracle/ref_shim): the lineplot module's PRESENT half (render transforms) uses glm, a network wrap of
// the reference's build that is absent here.  The compiled-reference checker never runs that half; these few types only let
// the translation unit that also holds the module's validate / define / create compile.  Not a glm replacement.
#pragma once
namespace glm {
struct vec2 { float x = 0, y = 0; vec2() = default; vec2(float a, float b) : x(a), y(b) {} };
struct vec3 { float x = 0, y = 0, z = 0; vec3() = default; vec3(float a, float b, float c) : x(a), y(b), z(c) {} };
struct mat4 {
    float m[16] = {0};
    mat4() = default;
    explicit mat4(float d) { m[0] = m[5] = m[10] = m[15] = d; }
};
inline mat4 translate(const mat4& a, const vec3& v) {
    mat4 r = a;
    for (int i = 0; i < 4; ++i) r.m[12 + i] = a.m[i] * v.x + a.m[4 + i] * v.y + a.m[8 + i] * v.z + a.m[12 + i];
    return r;
}
inline mat4 scale(const mat4& a, const vec3& v) {
    mat4 r = a;
    for (int i = 0; i < 4; ++i) { r.m[i] = a.m[i] * v.x; r.m[4 + i] = a.m[4 + i] * v.y; r.m[8 + i] = a.m[8 + i] * v.z; }
    return r;
}
}  // namespace glm
