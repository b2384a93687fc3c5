// host/camera.cpp — see camera.hpp.  Matrices are column-major float[16] like glm::mat4.
#include "camera.hpp"

#include <cmath>
#include <cstring>

namespace scene
{

namespace
{

struct V3
{
    float x, y, z;
};
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 norm3(V3 a)
{
    const float inv = 1.f / std::sqrt(dot3(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}

// Gauss-Jordan inverse of a column-major 4x4 (well conditioned for rigid/projective camera matrices)
void invert4(const float in[16], float out[16])
{
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
        {
            a[r][c] = in[c * 4 + r];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col)
    {
        int pivot = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[pivot][col])) pivot = r;
        if (pivot != col)
            for (int c = 0; c < 8; ++c)
            {
                const double t = a[col][c];
                a[col][c] = a[pivot][c];
                a[pivot][c] = t;
            }
        const double inv = 1.0 / a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] *= inv;
        for (int r = 0; r < 4; ++r)
        {
            if (r == col) continue;
            const double f = a[r][col];
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[c * 4 + r] = (float)a[r][4 + c];
}

void mul4(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row)
        {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a[k * 4 + row] * b[c * 4 + k];
            r[c * 4 + row] = s;
        }
    std::memcpy(out, r, sizeof(r));
}

} // namespace

void Camera::lookAt(const CameraTransform &transform)
{
    m_transform = transform;
    updateWorldToCamera();
}

void Camera::setParameters(const CameraParameters &parameters)
{
    m_parameters = parameters;
    m_changedThisFrame = true;
}

void Camera::updateResolution(uint32_t width, uint32_t height)
{
    if (m_resolution[0] != width || m_resolution[1] != height) m_changedThisFrame = true;
    m_resolution[0] = width;
    m_resolution[1] = height;
}

// src/scene/Camera.cpp:366-395
void Camera::updateWorldToCamera()
{
    const V3 eye{m_transform.eye[0], m_transform.eye[1], m_transform.eye[2]};
    const V3 target{m_transform.target[0], m_transform.target[1], m_transform.target[2]};
    const V3 up{m_transform.up[0], m_transform.up[1], m_transform.up[2]};
    const V3 fwd = norm3(sub(target, eye));
    const V3 z{-fwd.x, -fwd.y, -fwd.z};
    const V3 right = norm3(cross3(up, z));
    const V3 newUp = norm3(cross3(z, right));
    const float w2c[16] = {right.x, newUp.x, z.x, 0.f, right.y, newUp.y, z.y, 0.f,
                           right.z, newUp.z, z.z, 0.f, -dot3(right, eye), -dot3(newUp, eye), -dot3(z, eye), 1.f};
    std::memcpy(m_worldToCamera, w2c, sizeof(w2c));
    invert4(m_worldToCamera, m_cameraToWorld);
    m_maxViewScale = 1.f; // rows of a rigid look-at matrix have unit length
    m_changedThisFrame = true;
}

// src/scene/Camera.cpp:105-153
void Camera::perspective()
{
    const float ar = (float)m_resolution[0] / (float)m_resolution[1];
    // reverse-z: near and far swapped
    const float zN = m_parameters.zF;
    const float zF = m_parameters.zN;
    const float tf = 1.f / std::tan(m_parameters.fov * 0.5f);
    const float flip[16] = {1.f, 0.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 0.f, 0.5f, 0.f, 0.f, 0.f, 0.5f, 1.f};
    const float proj[16] = {tf / ar, 0.f, 0.f, 0.f, 0.f, tf, 0.f, 0.f,
                            0.f, 0.f, (zF + zN) / (zN - zF), -1.f, 0.f, 0.f, 2 * zF * zN / (zN - zF), 0.f};
    mul4(flip, proj, m_cameraToClip);
    float c2cw[16];
    mul4(m_cameraToClip, m_worldToCamera, c2cw);
    invert4(c2cw, m_clipToWorld);
    const float sensorHeight = sensorWidth() / ar;
    m_parameters.focalLength = sensorHeight * tf * 0.5f;
}

// src/scene/Camera.cpp:162-204
const prosper_CameraUniforms &Camera::updateBuffer()
{
    perspective();
    prosper_CameraUniforms u = {};
    std::memcpy(&u.worldToCamera, m_worldToCamera, 64);
    std::memcpy(&u.cameraToWorld, m_cameraToWorld, 64);
    std::memcpy(&u.cameraToClip, m_cameraToClip, 64);
    std::memcpy(&u.clipToWorld, m_clipToWorld, 64);
    std::memcpy(&u.previousWorldToCamera, m_worldToCamera, 64);
    std::memcpy(&u.previousCameraToClip, m_cameraToClip, 64);
    u.eye = prosper_vec4{m_transform.eye[0], m_transform.eye[1], m_transform.eye[2], 1.f};
    u.resolution[0] = m_resolution[0];
    u.resolution[1] = m_resolution[1];
    u.near_ = m_parameters.zN;
    u.far_ = m_parameters.zF;
    u.maxViewScale = m_maxViewScale;
    m_uniforms = u;
    return m_uniforms;
}

} // namespace scene
