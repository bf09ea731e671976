// arena_body.h — the slice of btRigidBody the tick needs (BulletDynamics/Dynamics/btRigidBody.{h,cpp}).
#pragma once
#include "arena_types.h"

namespace rlg {

RLG_HD void body_apply_central_impulse(Body& b, V3 imp, float inv_mass) { b.vel += imp * inv_mass; }
RLG_HD void body_apply_impulse(Body& b, V3 imp, V3 rel, float inv_mass) {  // btRigidBody.h:342-352
    b.vel += imp * inv_mass;
    b.angvel += b.inv_inertia_w * cross(rel, imp);
}
RLG_HD V3 body_vel_at(const Body& b, V3 rel) { return b.vel + cross(b.angvel, rel); }
// btRigidBody::computeImpulseDenominator (btRigidBody.h)
RLG_HD float body_impulse_denom(const Body& b, V3 pos, V3 n, float inv_mass) {
    V3 r0 = pos - b.pos;
    V3 c0 = cross(r0, n);
    V3 vec = cross(tmul(b.inv_inertia_w, c0), r0);
    return inv_mass + dot(n, vec);
}
// updateInertiaTensor (btRigidBody.cpp:252-255)
RLG_HD void body_update_inertia(Body& b, V3 inv_inertia_local) {
    b.inv_inertia_w = scaled_cols(b.rot, inv_inertia_local) * transpose(b.rot);
}
// world inertia tensor: the reference takes m_invInertiaTensorWorld.inverse() (Car.cpp:600,636,832);
// R diag(I) R^T is the same matrix up to rounding.
RLG_HD M3 body_inertia_w(const Body& b, V3 inertia_local) { return scaled_cols(b.rot, inertia_local) * transpose(b.rot); }

// The hitbox as Bullet really sees it (btBoxShape.cpp:17-27): the implicit core is halfExtents - 0.04 (the default
// margin), then setSafeMargin() LOWERS the margin to 0.1 * min(halfExtents) = 0.0386591 for the Octane, so the
// box "with margin" is 0.0013 BT smaller than the configured hitbox in every direction.
constexpr float BOX_MARGIN = 0.1f * ((K::HITBOX_Z * UU2BT) / 2);
RLG_HD V3 hitbox_core() { return v3((K::HITBOX_X * UU2BT) / 2 - 0.04f, (K::HITBOX_Y * UU2BT) / 2 - 0.04f, (K::HITBOX_Z * UU2BT) / 2 - 0.04f); }
RLG_HD V3 hitbox_half() { V3 c = hitbox_core(); return v3(c.x + BOX_MARGIN, c.y + BOX_MARGIN, c.z + BOX_MARGIN); }
constexpr float HITBOX_REACH = 1.6f;   // > |hitbox_half()| = 1.534 BT: no point of the hitbox (margin included) is farther from the hitbox centre, along any direction
RLG_HD V3 hitbox_off() { return v3(K::HITBOX_OFF_X, K::HITBOX_OFF_Y, K::HITBOX_OFF_Z) * UU2BT; }

RLG_HD V3 car_inertia_local() {  // btBoxShape::calculateLocalInertia (btBoxShape.cpp:34-47), mass 180
    V3 he = hitbox_half();
    float hx = he.x, hy = he.y, hz = he.z;
    float lx = 2.f * hx, ly = 2.f * hy, lz = 2.f * hz;
    return v3(K::CAR_MASS / 12.0f * (ly * ly + lz * lz), K::CAR_MASS / 12.0f * (lx * lx + lz * lz), K::CAR_MASS / 12.0f * (lx * lx + ly * ly));
}
RLG_HD V3 car_inv_inertia_local() { V3 i = car_inertia_local(); return v3(1.0f / i.x, 1.0f / i.y, 1.0f / i.z); }
RLG_HD V3 ball_inv_inertia_local() {  // btSphereShape::calculateLocalInertia (btSphereShape.cpp:66-70)
    float r = K::BALL_RADIUS * UU2BT;
    float e = 0.4f * K::BALL_MASS * r * r;
    return v3(1.0f / e, 1.0f / e, 1.0f / e);
}
constexpr float CAR_INV_MASS = 1.0f / K::CAR_MASS;
constexpr float BALL_INV_MASS = 1.0f / K::BALL_MASS;

// ---- wheel geometry of the Octane (Car.cpp:243-258) ----------------------------------------------------
RLG_HD V3 wheel_conn(int i) {  // Car.cpp:243-253
    bool front = i < 2, left = (i % 2) != 0;
    V3 o = front ? v3(K::WHEEL_FX, K::WHEEL_FY, K::WHEEL_FZ) : v3(K::WHEEL_BX, K::WHEEL_BY, K::WHEEL_BZ);
    if (left) o.y *= -1.f;
    return o * UU2BT;
}
RLG_HD float wheel_rest(int i) { return ((i < 2 ? K::SUS_REST_FRONT : K::SUS_REST_BACK) - K::MAX_SUSPENSION_TRAVEL) * UU2BT; }  // Car.cpp:255-258
RLG_HD float wheel_radius(int i) { return (i < 2 ? K::WHEEL_RAD_FRONT : K::WHEEL_RAD_BACK) * UU2BT; }
RLG_HD float wheel_travel() { return ((K::MAX_SUSPENSION_TRAVEL * UU2BT) * 100) / 100; }  // m_maxSuspensionTravelCm / 100

}  // namespace rlg
