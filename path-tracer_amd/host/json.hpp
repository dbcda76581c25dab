// json.hpp -- minimal JSON reader for glTF 2.0 files (objects, arrays, strings, numbers, booleans, null).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::map<std::string, Json> obj;

    bool has(const std::string& k) const { return type == Object && obj.count(k) != 0; }
    const Json& operator[](const std::string& k) const
    {
        static const Json none;
        auto it = obj.find(k);
        return it == obj.end() ? none : it->second;
    }
    const Json& operator[](size_t i) const { return arr.at(i); }
    size_t size() const { return type == Array ? arr.size() : obj.size(); }
    double number(double fallback) const { return type == Number ? num : fallback; }

    static Json parse(const std::string& text)
    {
        size_t p = 0;
        Json j = value(text, p);
        return j;
    }

private:
    static void ws(const std::string& s, size_t& p) { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\r' || s[p] == '\t')) ++p; }
    static Json value(const std::string& s, size_t& p)
    {
        ws(s, p);
        if (p >= s.size()) throw std::runtime_error("json: unexpected end");
        Json j;
        char c = s[p];
        if (c == '{') {
            j.type = Object; ++p; ws(s, p);
            if (s[p] == '}') { ++p; return j; }
            for (;;) {
                ws(s, p);
                Json k = value(s, p);
                ws(s, p);
                if (s[p] != ':') throw std::runtime_error("json: expected ':'");
                ++p;
                j.obj[k.str] = value(s, p);
                ws(s, p);
                if (s[p] == ',') { ++p; continue; }
                if (s[p] == '}') { ++p; break; }
                throw std::runtime_error("json: expected ',' or '}'");
            }
        } else if (c == '[') {
            j.type = Array; ++p; ws(s, p);
            if (s[p] == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(value(s, p));
                ws(s, p);
                if (s[p] == ',') { ++p; continue; }
                if (s[p] == ']') { ++p; break; }
                throw std::runtime_error("json: expected ',' or ']'");
            }
        } else if (c == '"') {
            j.type = String; ++p;
            while (p < s.size() && s[p] != '"') {
                if (s[p] == '\\' && p + 1 < s.size()) {
                    char e = s[p + 1];
                    if (e == 'n') j.str += '\n'; else if (e == 't') j.str += '\t'; else if (e == 'u') { j.str += '?'; p += 4; } else j.str += e;
                    p += 2;
                } else j.str += s[p++];
            }
            ++p;
        } else if (s.compare(p, 4, "true") == 0) { j.type = Bool; j.b = true; p += 4; }
        else if (s.compare(p, 5, "false") == 0) { j.type = Bool; j.b = false; p += 5; }
        else if (s.compare(p, 4, "null") == 0) { p += 4; }
        else {
            char* end = nullptr;
            j.type = Number;
            j.num = std::strtod(s.c_str() + p, &end);
            if (end == s.c_str() + p) throw std::runtime_error("json: bad token");
            p = static_cast<size_t>(end - s.c_str());
        }
        return j;
    }
};
