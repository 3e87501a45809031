// ini.hpp -- settings.ini reader with the semantics the reference gets from inih's INIReader
// (libs/inih/INIReader.h:312-461): [section] headers, name=value or name:value, ';' and '#' full-line comments,
// inline " ;" comments, case-insensitive section/name lookup, GetInteger/GetFloat/GetBoolean conversions,
// ParseError() < 0 when the file cannot be opened.  Unknown keys are stored and never read.
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>

class INIReader {
public:
    explicit INIReader(const std::string& filename)
    {
        std::ifstream f(filename);
        if (!f.is_open()) { error_ = -1; return; }
        std::string line, section;
        int lineno = 0;
        while (std::getline(f, line)) {
            ++lineno;
            if (lineno == 1 && line.size() >= 3 && (unsigned char)line[0] == 0xEF && (unsigned char)line[1] == 0xBB && (unsigned char)line[2] == 0xBF) line = line.substr(3);
            std::string s = trim(line);
            if (s.empty() || s[0] == ';' || s[0] == '#') continue;
            if (s[0] == '[') {
                const auto e = s.find(']');
                if (e == std::string::npos) { if (!error_) error_ = lineno; continue; }
                section = trim(s.substr(1, e - 1));
                continue;
            }
            auto sep = s.find_first_of("=:");
            if (sep == std::string::npos) { if (!error_) error_ = lineno; continue; }
            std::string name = trim(s.substr(0, sep));
            std::string value = s.substr(sep + 1);
            // inline comment: ';' preceded by whitespace (INIReader.h:76-77)
            for (size_t i = 1; i < value.size(); ++i)
                if (value[i] == ';' && std::isspace((unsigned char)value[i - 1])) { value = value.substr(0, i); break; }
            values_[key(section, name)] = trim(value);
        }
    }
    int ParseError() const { return error_; }
    std::string Get(const std::string& section, const std::string& name, const std::string& def) const
    {
        auto it = values_.find(key(section, name));
        return it == values_.end() ? def : it->second;
    }
    long GetInteger(const std::string& section, const std::string& name, long def) const
    {
        const std::string v = Get(section, name, "");
        char* end = nullptr;
        const long n = std::strtol(v.c_str(), &end, 0);
        return end > v.c_str() ? n : def;
    }
    float GetFloat(const std::string& section, const std::string& name, float def) const
    {
        const std::string v = Get(section, name, "");
        char* end = nullptr;
        const float n = std::strtof(v.c_str(), &end);
        return end > v.c_str() ? n : def;
    }
    bool GetBoolean(const std::string& section, const std::string& name, bool def) const
    {
        std::string v = Get(section, name, "");
        std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        if (v == "true" || v == "yes" || v == "on" || v == "1") return true;
        if (v == "false" || v == "no" || v == "off" || v == "0") return false;
        return def;
    }

private:
    int error_ = 0;
    std::map<std::string, std::string> values_;
    static std::string trim(const std::string& s)
    {
        size_t a = 0, b = s.size();
        while (a < b && std::isspace((unsigned char)s[a])) ++a;
        while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
        return s.substr(a, b - a);
    }
    static std::string key(const std::string& section, const std::string& name)
    {
        std::string k = section + "=" + name;
        std::transform(k.begin(), k.end(), k.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        return k;
    }
};
