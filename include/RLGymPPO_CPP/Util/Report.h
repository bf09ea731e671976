// Report (PUB/Util/Report.h:5-109): named double metrics with sum / average accumulation
#pragma once
#include "../Framework.h"
namespace RLGPC {
struct Report {
    typedef double Val;
    std::map<std::string, Val> data;
    Val& operator[](const std::string& key) { return data[key]; }
    Val operator[](const std::string& key) const { return data.at(key); }
    bool Has(const std::string& key) const { return data.count(key) != 0; }
    void Accum(const std::string& key, Val val) { data[key] += val; }   // a missing key starts at 0
    void AccumAvg(const std::string& key, Val val) { Accum(key + "_avg_total", val); Accum(key + "_avg_count", 1); }
    Val GetAvg(const std::string& key) const {
        Val n = data.at(key + "_avg_count");
        return n > 0 ? data.at(key + "_avg_total") / n : 0;
    }
    std::string SingleToString(const std::string& key, bool digitCommas = false) const {
        std::ostringstream out;
        Val v = data.at(key);
        out << key << ": ";
        double a = std::fabs(v);
        if ((a < 1e-3 && v != 0) || a >= 1e11) out << std::scientific << v;
        else if (v == (double)(int64_t)v) {
            std::string digits = std::to_string((int64_t)std::llabs((long long)v)), grouped;
            for (size_t i = 0; i < digits.size(); i++) { if (digitCommas && i && (digits.size() - i) % 3 == 0) grouped += ','; grouped += digits[i]; }
            out << (v < 0 ? "-" : "") << grouped;
        } else out << std::fixed << std::setprecision(4) << v;
        return out.str();
    }
    std::string ToString(bool digitCommas = false, const std::string& prefix = {}) const {
        std::string s;
        for (auto& kv : data) s += prefix + SingleToString(kv.first, digitCommas) + "\n";
        return s;
    }
    void Display(const std::vector<std::string>& keyRows) const {   // empty string = blank line, leading '-' = indent
        for (std::string row : keyRows) {
            if (row.empty()) { RG_LOG(""); continue; }
            int indent = 0; while (!row.empty() && row[0] == '-') { indent++; row.erase(0, 1); }
            if (Has(row)) RG_LOG(std::string(indent * 2, ' ') << SingleToString(row, true));
        }
    }
    void Clear() { data.clear(); }
    Report operator+(const Report& o) const { Report r = *this; r.data.insert(o.data.begin(), o.data.end()); return r; }
    Report& operator+=(const Report& o) { data.insert(o.data.begin(), o.data.end()); return *this; }
};
}
