// host_util.h -- error reporting and the small INI reader (no boost) of libddrl_pop.so.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

// The reference reports errors with LOG_ERROR + exit(1) (MLP.hpp:99-100); here every entry point
// returns a code and leaves the text in a thread-local buffer (ddrl_last_error()).
int ddrl_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Reads the reference's .ini files: boost::property_tree::ini_parser syntax ([section], key=value,
// ';' comments), keys addressed as "section.key" (common/include/arch/Simulator.hpp:300-328),
// list values "a:b:c" or "a,b,c" as bib::to_array (common/include/bib/IniParser.hpp:11-24).
class DdrlIni {
 public:
  bool load(const char *path, std::string *err) {
    std::ifstream f(path);
    if (!f) { *err = std::string("cannot open ini file '") + path + "'"; return false; }
    std::string line, section;
    int ln = 0;
    while (std::getline(f, line)) {
      ++ln;
      std::string s = trim(line);
      if (s.empty() || s[0] == ';' || s[0] == '#') continue;
      if (s[0] == '[') {
        size_t e = s.find(']');
        if (e == std::string::npos) { *err = std::string(path) + ":" + std::to_string(ln) + ": unmatched '['"; return false; }
        section = trim(s.substr(1, e - 1));
        continue;
      }
      size_t eq = s.find('=');
      if (eq == std::string::npos) { *err = std::string(path) + ":" + std::to_string(ln) + ": '=' expected"; return false; }
      std::string k = trim(s.substr(0, eq)), v = trim(s.substr(eq + 1));
      kv_[section.empty() ? k : section + "." + k] = v;
    }
    return true;
  }
  bool has(const std::string &k) const { return kv_.count(k) != 0; }
  const std::string &get(const std::string &k) const {
    auto it = kv_.find(k);
    if (it == kv_.end()) throw std::runtime_error("No such node (" + k + ")");   // ptree_bad_path text
    return it->second;
  }
  double get_double(const std::string &k) const {
    const std::string &v = get(k);
    char *end = nullptr;
    double d = strtod(v.c_str(), &end);
    if (end == v.c_str() || *end) throw std::runtime_error("conversion of data to type \"double\" failed (" + k + "=" + v + ")");
    return d;
  }
  long get_long(const std::string &k) const {
    const std::string &v = get(k);
    char *end = nullptr;
    long d = strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end) throw std::runtime_error("conversion of data to type \"int\" failed (" + k + "=" + v + ")");
    return d;
  }
  int get_bool(const std::string &k) const {   // boost stream_translator<bool>: 0/1 or true/false
    const std::string &v = get(k);
    if (v == "1" || v == "true") return 1;
    if (v == "0" || v == "false") return 0;
    throw std::runtime_error("conversion of data to type \"bool\" failed (" + k + "=" + v + ")");
  }
  std::vector<long> get_list(const std::string &k) const {
    std::vector<long> out;
    const std::string &v = get(k);
    std::string tok;
    for (size_t i = 0; i <= v.size(); ++i) {
      if (i == v.size() || v[i] == ':' || v[i] == ',') {
        tok = trim(tok);
        if (!tok.empty()) out.push_back(strtol(tok.c_str(), nullptr, 10));
        tok.clear();
      } else tok += v[i];
    }
    return out;
  }

 private:
  static std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  }
  std::map<std::string, std::string> kv_;
};
