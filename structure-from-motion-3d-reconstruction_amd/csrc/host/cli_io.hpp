// cli_io.hpp — the reference CLI's on-disk surface: config.json (cpp/include/minijson.hpp),
// Middlebury *_par.txt / *_ang.txt (T:111-152) and binary PGM (cpp/include/pgm_io.hpp:24-54).
// Error strings follow the reference so that `ERROR: ...` lines match.
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_math.hpp"

namespace sfmx_cli {

// ------------------------------------------------------------------------------------------ JSON
struct Json {
  enum class T { Null, Bool, Num, Str, Obj, Arr } type = T::Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<std::pair<std::string, Json>> obj;  // first occurrence of a key wins (emplace semantics)
  std::vector<Json> arr;
  const Json* get(const std::string& k) const {
    if (type != T::Obj) return nullptr;
    for (const auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

class JsonParser {
 public:
  explicit JsonParser(const std::string& s) : p_(s.c_str()), end_(s.c_str() + s.size()) {}
  Json parse() {
    ws();
    Json v = value();
    ws();
    if (p_ != end_) throw std::runtime_error("Trailing characters after JSON");
    return v;
  }

 private:
  const char* p_;
  const char* end_;
  [[noreturn]] void err(const std::string& m) { throw std::runtime_error("JSON parse error: " + m); }
  void ws() { while (p_ < end_ && std::isspace((unsigned char)*p_)) ++p_; }
  bool match(char c) { if (p_ < end_ && *p_ == c) { ++p_; return true; } return false; }
  void expect(char c) { if (!match(c)) err(std::string("Expected '") + c + "'"); }
  bool word(const char* w) {
    const size_t n = std::char_traits<char>::length(w);
    if ((size_t)(end_ - p_) >= n && std::string(p_, n) == w) { p_ += n; return true; }
    return false;
  }
  Json value() {
    ws();
    if (p_ >= end_) err("Unexpected end of input");
    const char c = *p_;
    Json v;
    if (c == 'n') { if (!word("null")) err("Invalid token (expected null)"); return v; }
    if (c == 't' || c == 'f') {
      v.type = Json::T::Bool;
      if (word("true")) v.b = true;
      else if (word("false")) v.b = false;
      else err("Invalid token (expected true/false)");
      return v;
    }
    if (c == '"') { v.type = Json::T::Str; v.str = string(); return v; }
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '-' || std::isdigit((unsigned char)c)) { v.type = Json::T::Num; v.num = number(); return v; }
    err(std::string("Unexpected character '") + c + "'");
  }
  static int hex(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return 10 + (c - 'a');
    if (c >= 'A' && c <= 'F') return 10 + (c - 'A');
    return -1;
  }
  std::string string() {
    expect('"');
    std::string out;
    while (p_ < end_) {
      const char c = *p_++;
      if (c == '"') return out;
      if (c != '\\') { out.push_back(c); continue; }
      if (p_ >= end_) err("Bad escape");
      const char e = *p_++;
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          if (end_ - p_ < 4) err("Bad \\u escape");
          int v = 0;
          for (int i = 0; i < 4; i++) {
            const int h = hex(p_[i]);
            if (h < 0) err("Bad hex in \\u escape");
            v = (v << 4) | h;
          }
          p_ += 4;
          if (v <= 0x7F) out.push_back((char)v);
          else if (v <= 0x7FF) { out.push_back((char)(0xC0 | ((v >> 6) & 0x1F))); out.push_back((char)(0x80 | (v & 0x3F))); }
          else { out.push_back((char)(0xE0 | ((v >> 12) & 0x0F))); out.push_back((char)(0x80 | ((v >> 6) & 0x3F))); out.push_back((char)(0x80 | (v & 0x3F))); }
          break;
        }
        default: err("Unknown escape");
      }
    }
    err("Unterminated string");
  }
  double number() {
    const char* start = p_;
    (void)match('-');
    if (p_ >= end_) err("Bad number");
    if (*p_ == '0') ++p_;
    else {
      if (!std::isdigit((unsigned char)*p_)) err("Bad number");
      while (p_ < end_ && std::isdigit((unsigned char)*p_)) ++p_;
    }
    if (p_ < end_ && *p_ == '.') {
      ++p_;
      if (p_ >= end_ || !std::isdigit((unsigned char)*p_)) err("Bad fraction");
      while (p_ < end_ && std::isdigit((unsigned char)*p_)) ++p_;
    }
    if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
      ++p_;
      if (p_ < end_ && (*p_ == '+' || *p_ == '-')) ++p_;
      if (p_ >= end_ || !std::isdigit((unsigned char)*p_)) err("Bad exponent");
      while (p_ < end_ && std::isdigit((unsigned char)*p_)) ++p_;
    }
    const std::string tmp(start, p_);
    char* ep = nullptr;
    const double v = std::strtod(tmp.c_str(), &ep);
    if (ep == tmp.c_str()) err("Bad number conversion");
    return v;
  }
  Json array() {
    expect('[');
    Json out;
    out.type = Json::T::Arr;
    ws();
    if (match(']')) return out;
    while (true) {
      out.arr.push_back(value());
      ws();
      if (match(']')) break;
      expect(',');
      ws();
    }
    return out;
  }
  Json object() {
    expect('{');
    Json out;
    out.type = Json::T::Obj;
    ws();
    if (match('}')) return out;
    while (true) {
      if (p_ >= end_ || *p_ != '"') err("Expected string key");
      std::string key = string();
      ws();
      expect(':');
      ws();
      Json v = value();
      if (!out.get(key)) out.obj.emplace_back(std::move(key), std::move(v));
      ws();
      if (match('}')) break;
      expect(',');
      ws();
    }
    return out;
  }
};

// T:65-106: cpp.* overrides common.*
inline const Json* jget(const Json& v, std::initializer_list<const char*> path) {
  const Json* cur = &v;
  for (const char* k : path) {
    cur = cur->get(k);
    if (!cur) return nullptr;
  }
  return cur;
}
inline const Json* jpick(const Json& root, const char* sec, const char* key) {
  if (const Json* a = jget(root, {"cpp", sec, key})) return a;
  return jget(root, {"common", sec, key});
}
inline std::optional<int> jint(const Json* v) {
  if (v && v->type == Json::T::Num) return (int)std::llround(v->num);
  return std::nullopt;
}
inline std::optional<double> jdouble(const Json* v) {
  if (v && v->type == Json::T::Num) return v->num;
  return std::nullopt;
}
inline std::optional<std::string> jstring(const Json* v) {
  if (v && v->type == Json::T::Str) return v->str;
  return std::nullopt;
}

// ------------------------------------------------------------------------------------------ dataset files
struct MBRecord {
  std::string img;
  sfmx_host::Mat3 K, Rwc;
  sfmx_host::V3 twc;
};
// T:120-140
inline std::vector<MBRecord> read_par(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Failed to open: " + path);
  int n = 0;
  f >> n;
  std::vector<MBRecord> recs;
  recs.reserve((size_t)std::max(0, n));
  for (int i = 0; i < n; i++) {
    MBRecord r;
    f >> r.img;
    double v[21] = {0};
    for (double& x : v) f >> x;
    for (int k = 0; k < 9; k++) { r.K.a[k] = v[k]; r.Rwc.a[k] = v[9 + k]; }
    r.twc = {v[18], v[19], v[20]};
    recs.push_back(r);
  }
  return recs;
}
struct MBAngle { double lat = 0, lon = 0; };
// T:142-152 (first occurrence of a name wins, as with unordered_map::emplace)
inline std::unordered_map<std::string, MBAngle> read_ang(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Failed to open: " + path);
  std::unordered_map<std::string, MBAngle> a;
  std::string img;
  double lat, lon;
  while (f >> lat >> lon >> img) a.emplace(img, MBAngle{lat, lon});
  return a;
}

struct Gray {
  int w = 0, h = 0;
  std::vector<std::uint8_t> pix;
};
// cpp/include/pgm_io.hpp:24-54
inline Gray read_pgm(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Failed to open: " + path);
  auto skip_comments = [&]() {
    while (f.peek() == '#') {
      std::string line;
      std::getline(f, line);
    }
  };
  std::string magic;
  f >> magic;
  if (magic != "P5") throw std::runtime_error("Only binary PGM (P5) supported: " + path);
  skip_comments();
  int w = 0, h = 0, maxv = 0;
  f >> w;
  skip_comments();
  f >> h;
  skip_comments();
  f >> maxv;
  if (maxv != 255) throw std::runtime_error("Only 8-bit PGM supported: " + path);
  f.get();
  Gray im;
  im.w = w;
  im.h = h;
  im.pix.resize((size_t)w * (size_t)h);
  f.read(reinterpret_cast<char*>(im.pix.data()), (std::streamsize)im.pix.size());
  if (!f) throw std::runtime_error("PGM read failed: " + path);
  return im;
}

}  // namespace sfmx_cli
