// RenderSender (PUB/Util/RenderSender.{h,cpp} + python_scripts/render_receiver.py): the reference serialises the rendered game's
// state to JSON, hands it to an embedded Python module, and that module forwards a RocketSimVis datagram over UDP
// (127.0.0.1:9273).  Here the datagram is built and sent natively, with the schema the receiver produces:
//   {"gamemode", "ball_phys": {ang_vel, pos, vel}, "cars": [{ball_touched, boost_amount, boost_pickups, car_id, has_flip, is_demoed, on_ground,
//    phys{ang_vel, forward, pos, right, up, vel}, team_num}], "boost_pad_states": [34 x bool]}   -- byte for byte (tests/golden/sender_golden.json)
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>
#include <algorithm>
#include <charconv>
#include <cmath>
#include <string>
#include <RLGymSim_CPP/Utils/Gamestates/GameState.h>
#include "../Framework.h"
namespace RLGPC {
struct RenderSender {
    int sock = -1;
    sockaddr_in addr{};
    uint64_t sent = 0;

    RenderSender(const char* ip = "127.0.0.1", int port = 9273) {
        RG_LOG("Initializing RenderSender...");
        sock = socket(AF_INET, SOCK_DGRAM, 0);
        if (sock < 0) RG_ERR_CLOSE("RenderSender: Failed to create the UDP socket");
        addr.sin_family = AF_INET; addr.sin_port = htons((uint16_t)port);
        if (inet_pton(AF_INET, ip, &addr.sin_addr) != 1) RG_ERR_CLOSE("RenderSender: bad address " << ip);
        RG_LOG(" > RenderSender initalized (RocketSimVis datagrams to " << ip << ":" << port << ").");
    }
    RenderSender(const RenderSender&) = delete;
    RenderSender& operator=(const RenderSender&) = delete;
    ~RenderSender() { if (sock >= 0) close(sock); }

    // A float as the receiver's json.dumps prints it: the float widened to double (nlohmann stores number_float as double, RenderSender.cpp:22-24)
    // in Python's repr -- shortest digits that round-trip, fixed notation for 1e-4 <= |x| < 1e16 with ".0" on whole numbers, else d.ddde+XX
    static std::string PyFloat(float f) {
        const double d = (double)f;
        if (std::isnan(d)) return "NaN";
        if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
        if (d == 0) return std::signbit(d) ? "-0.0" : "0.0";
        char buf[64];
        auto r = std::to_chars(buf, buf + sizeof buf, d, std::chars_format::scientific);   // shortest round-trip digits: [-]d[.ddd]e[+-]XX
        std::string sci(buf, r.ptr);
        const bool neg = sci[0] == '-';
        if (neg) sci.erase(0, 1);
        const size_t e = sci.find('e');
        std::string digits = sci.substr(0, e); digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
        const int exp10 = std::atoi(sci.c_str() + e + 1), decpt = exp10 + 1, nd = (int)digits.size();
        std::string out;
        if (decpt > -4 && decpt <= 16) {
            if (decpt <= 0) out = "0." + std::string((size_t)-decpt, '0') + digits;
            else if (decpt >= nd) out = digits + std::string((size_t)(decpt - nd), '0') + ".0";
            else out = digits.substr(0, (size_t)decpt) + "." + digits.substr((size_t)decpt);
        } else {
            out = digits.substr(0, 1) + (nd > 1 ? "." + digits.substr(1) : "") + "e" + (exp10 < 0 ? "-" : "+");
            const int ae = exp10 < 0 ? -exp10 : exp10;
            out += (ae < 10 ? "0" : "") + std::to_string(ae);
        }
        return (neg ? "-" : "") + out;
    }
    static void VecJson(std::ostream& o, const char* key, const RLGSC::Vec& v) { o << "\"" << key << "\": [" << PyFloat(v.x) << ", " << PyFloat(v.y) << ", " << PyFloat(v.z) << "]"; }

    // The datagram for one state, byte for byte what the reference sends: RenderSender.cpp:26-92 serialises through nlohmann::json, whose objects are
    // std::maps (keys SORTED); render_receiver.py:17-31 parses that, builds {gamemode, ball_phys (the ball without forward / right / up), cars (the
    // players as they are), boost_pad_states} in this order and sends json.dumps of it (", " and ": " separators, Python float repr).  The actions
    // and team_goals never leave the receiver.  Pinned by tests/golden/sender_golden.json, recorded from the reference's own render_receiver.py.
    static std::string ToJSON(const RLGSC::GameState& state, const RLGSC::ActionSet& actions) {
        (void)actions;
        std::ostringstream o;
        o << "{\"gamemode\": \"soccar\", \"ball_phys\": {";
        VecJson(o, "ang_vel", state.ball.angVel); o << ", "; VecJson(o, "pos", state.ball.pos); o << ", "; VecJson(o, "vel", state.ball.vel);
        o << "}, \"cars\": [";
        for (size_t i = 0; i < state.players.size(); i++) {
            const RLGSC::PlayerData& p = state.players[i];
            o << (i ? ", " : "") << "{\"ball_touched\": " << (p.ballTouchedStep ? "true" : "false") << ", \"boost_amount\": " << PyFloat(p.boostFraction)
              << ", \"boost_pickups\": " << p.boostPickups << ", \"car_id\": " << p.carId << ", \"has_flip\": " << (p.hasFlip ? "true" : "false")
              << ", \"is_demoed\": " << (p.carState.isDemoed ? "true" : "false") << ", \"on_ground\": " << (p.carState.isOnGround ? "true" : "false") << ", \"phys\": {";
            VecJson(o, "ang_vel", p.phys.angVel); o << ", "; VecJson(o, "forward", p.phys.rotMat.forward); o << ", "; VecJson(o, "pos", p.phys.pos); o << ", ";
            VecJson(o, "right", p.phys.rotMat.right); o << ", "; VecJson(o, "up", p.phys.rotMat.up); o << ", "; VecJson(o, "vel", p.phys.vel);
            o << "}, \"team_num\": " << (int)p.team << "}";
        }
        o << "], \"boost_pad_states\": [";
        for (int i = 0; i < RLGSC::CommonValues::BOOST_LOCATIONS_AMOUNT; i++) o << (i ? ", " : "") << (state.boostPads[i] ? "true" : "false");
        o << "]}";
        return o.str();
    }

    void Send(const RLGSC::GameState& state, const RLGSC::ActionSet& actions) {
        const std::string j = ToJSON(state, actions);
        // like the Python receiver's sendto: fire and forget (no viewer listening is not an error)
        (void)sendto(sock, j.data(), j.size(), 0, (const sockaddr*)&addr, sizeof(addr));
        sent++;
    }
};
}
