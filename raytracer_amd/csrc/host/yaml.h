// A reader for the YAML subset the scene files use: block sequences and
// mappings by indentation, single-line flow sequences / mappings (including a
// bare `key: value` pair inside a flow sequence), quoted and plain scalars and
// `#` comments. It stands in for serde_yaml on the host side (reference
// src/scene/mod.rs:71 `from_reader`), which only ever sees these constructs in
// scenes/*.yaml.
#pragma once

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {

struct YVal {
    enum Kind { Null, Scalar, Seq, Map } kind = Null;
    std::string s;          // scalar text
    bool quoted = false;    // scalar came from "..." or '...'
    std::vector<YVal> seq;
    std::vector<std::pair<std::string, YVal>> map;

    bool is_map() const { return kind == Map; }
    bool is_seq() const { return kind == Seq; }
    bool is_scalar() const { return kind == Scalar; }
    const YVal* get(const std::string& key) const {
        if (kind != Map) return nullptr;
        for (auto& kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return get(key) != nullptr; }
    void set(const std::string& key, YVal v) {
        kind = Map;
        for (auto& kv : map)
            if (kv.first == key) { kv.second = std::move(v); return; }
        map.emplace_back(key, std::move(v));
    }
    bool remove(const std::string& key, YVal* out = nullptr) {
        for (size_t i = 0; i < map.size(); i++)
            if (map[i].first == key) {
                if (out) *out = std::move(map[i].second);
                map.erase(map.begin() + i);
                return true;
            }
        return false;
    }
    // A plain (unquoted) scalar that parses fully as a number.
    bool is_number() const {
        if (kind != Scalar || quoted || s.empty()) return false;
        char* end = nullptr;
        std::strtod(s.c_str(), &end);
        return end && *end == '\0';
    }
    double as_f64() const {
        if (!is_number()) throw std::runtime_error("Failed to parse '" + s + "' as an f64");
        return std::strtod(s.c_str(), nullptr);
    }
    uint64_t as_u64() const {
        if (kind != Scalar || quoted) throw std::runtime_error("expected an integer, found '" + s + "'");
        char* end = nullptr;
        unsigned long long v = std::strtoull(s.c_str(), &end, 10);
        if (!end || *end != '\0' || s.empty() || s[0] == '-')
            throw std::runtime_error("expected an unsigned integer, found '" + s + "'");
        return v;
    }
    bool as_bool() const {
        if (kind == Scalar && !quoted) {
            if (s == "true") return true;
            if (s == "false") return false;
        }
        throw std::runtime_error("expected a bool, found '" + s + "'");
    }
    // A string scalar: quoted, or plain and not a number / bool.
    bool is_string() const { return kind == Scalar && (quoted || !is_number()); }
};

namespace yaml_detail {

struct Line { int indent; std::string text; int number; };

inline std::string rtrim(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}
inline std::string trim(const std::string& s) {
    size_t b = 0;
    while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) b++;
    return rtrim(s.substr(b));
}
// Strip a `#` comment that is at line start or preceded by whitespace and not
// inside quotes.
inline std::string strip_comment(const std::string& s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (q) { if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') { q = c; continue; }
        if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;

    [[noreturn]] void fail(const std::string& msg, int line) {
        throw std::runtime_error("yaml line " + std::to_string(line) + ": " + msg);
    }

    // ---- flow / scalar parsing on a single string
    static YVal scalar(const std::string& raw) {
        YVal v;
        std::string t = trim(raw);
        v.kind = YVal::Scalar;
        if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\''))) {
            v.quoted = true;
            v.s = t.substr(1, t.size() - 2);
        } else if (t.empty() || t == "~" || t == "null") {
            v.kind = YVal::Null;
        } else {
            v.s = t;
        }
        return v;
    }
    // Find `: ` (or trailing `:`) outside quotes / brackets; npos if none.
    static size_t find_key_colon(const std::string& s) {
        char q = 0;
        int depth = 0;
        for (size_t i = 0; i < s.size(); i++) {
            char c = s[i];
            if (q) { if (c == q) q = 0; continue; }
            if (c == '"' || c == '\'') { q = c; continue; }
            if (c == '[' || c == '{') depth++;
            if (c == ']' || c == '}') depth--;
            if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
        }
        return std::string::npos;
    }
    // Split the inside of [...] or {...} at top-level commas.
    static std::vector<std::string> split_flow(const std::string& s) {
        std::vector<std::string> out;
        char q = 0;
        int depth = 0;
        std::string cur;
        for (char c : s) {
            if (q) { cur += c; if (c == q) q = 0; continue; }
            if (c == '"' || c == '\'') { q = c; cur += c; continue; }
            if (c == '[' || c == '{') depth++;
            if (c == ']' || c == '}') depth--;
            if (c == ',' && depth == 0) { out.push_back(cur); cur.clear(); continue; }
            cur += c;
        }
        if (!trim(cur).empty() || !out.empty()) out.push_back(cur);
        return out;
    }
    YVal inline_value(const std::string& raw, int line) {
        std::string t = trim(raw);
        if (!t.empty() && t.front() == '[') {
            if (t.back() != ']') fail("multi-line flow sequences are not supported", line);
            YVal v;
            v.kind = YVal::Seq;
            for (auto& item : split_flow(t.substr(1, t.size() - 2))) {
                std::string it = trim(item);
                size_t c = (it.empty() || it.front() == '[' || it.front() == '{' || it.front() == '"')
                               ? std::string::npos : find_key_colon(it);
                if (c != std::string::npos) {  // `[rotate-y, degrees: 50]`
                    YVal m;
                    m.kind = YVal::Map;
                    m.map.emplace_back(trim(it.substr(0, c)), inline_value(it.substr(c + 1), line));
                    v.seq.push_back(std::move(m));
                } else {
                    v.seq.push_back(inline_value(it, line));
                }
            }
            return v;
        }
        if (!t.empty() && t.front() == '{') {
            if (t.back() != '}') fail("multi-line flow mappings are not supported", line);
            YVal v;
            v.kind = YVal::Map;
            for (auto& item : split_flow(t.substr(1, t.size() - 2))) {
                std::string it = trim(item);
                size_t c = find_key_colon(it);
                if (c == std::string::npos) fail("expected key: value in flow mapping", line);
                v.map.emplace_back(scalar(it.substr(0, c)).s, inline_value(it.substr(c + 1), line));
            }
            return v;
        }
        return scalar(t);
    }

    // ---- block parsing
    YVal parse_node(int indent) {
        if (pos >= lines.size()) return YVal();
        const Line& l = lines[pos];
        if (l.text == "-" || l.text.rfind("- ", 0) == 0) return parse_seq(l.indent);
        if (find_key_colon(l.text) != std::string::npos && l.text.front() != '[' && l.text.front() != '{' &&
            l.text.front() != '"')
            return parse_map(l.indent);
        (void)indent;
        YVal v = inline_value(l.text, l.number);
        pos++;
        return v;
    }
    YVal parse_seq(int indent) {
        YVal v;
        v.kind = YVal::Seq;
        while (pos < lines.size() && lines[pos].indent == indent &&
               (lines[pos].text == "-" || lines[pos].text.rfind("- ", 0) == 0)) {
            Line& l = lines[pos];
            std::string rest = l.text.size() > 1 ? l.text.substr(1) : "";
            size_t lead = 0;
            while (lead < rest.size() && rest[lead] == ' ') lead++;
            rest = rest.substr(lead);
            if (rest.empty()) {
                pos++;
                if (pos < lines.size() && lines[pos].indent > indent) v.seq.push_back(parse_node(lines[pos].indent));
                else v.seq.push_back(YVal());
            } else {
                // Re-home the item's content at its own column and parse it as a node.
                l.indent = indent + 1 + (int)lead;
                l.text = rest;
                v.seq.push_back(parse_node(l.indent));
            }
        }
        return v;
    }
    YVal parse_map(int indent) {
        YVal v;
        v.kind = YVal::Map;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line& l = lines[pos];
            if (l.text == "-" || l.text.rfind("- ", 0) == 0) break;
            size_t c = find_key_colon(l.text);
            if (c == std::string::npos) fail("expected 'key: value'", l.number);
            std::string key = scalar(l.text.substr(0, c)).s;
            std::string rest = trim(l.text.substr(c + 1));
            int number = l.number;
            pos++;
            if (!rest.empty()) {
                v.map.emplace_back(key, inline_value(rest, number));
            } else if (pos < lines.size() && lines[pos].indent > indent) {
                v.map.emplace_back(key, parse_node(lines[pos].indent));
            } else if (pos < lines.size() && lines[pos].indent == indent &&
                       (lines[pos].text == "-" || lines[pos].text.rfind("- ", 0) == 0)) {
                v.map.emplace_back(key, parse_seq(indent));
            } else {
                v.map.emplace_back(key, YVal());
            }
        }
        return v;
    }
};

}  // namespace yaml_detail

inline YVal yaml_parse(const std::string& text) {
    yaml_detail::Parser p;
    std::istringstream in(text);
    std::string raw;
    int number = 0;
    while (std::getline(in, raw)) {
        number++;
        std::string s = yaml_detail::rtrim(yaml_detail::strip_comment(raw));
        size_t ind = 0;
        while (ind < s.size() && s[ind] == ' ') ind++;
        if (ind == s.size()) continue;
        if (s == "---") continue;
        p.lines.push_back({(int)ind, s.substr(ind), number});
    }
    if (p.lines.empty()) return YVal();
    YVal v = p.parse_node(p.lines[0].indent);
    if (p.pos != p.lines.size())
        throw std::runtime_error("yaml line " + std::to_string(p.lines[p.pos].number) + ": unexpected indentation");
    return v;
}

inline YVal yaml_parse_file(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return yaml_parse(ss.str());
}

}  // namespace rt
