#include "model.h"

#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ct2x {

size_t dtype_size(int dtype) {
  switch (dtype) {
    case kF32: return 4;
    case kI8: return 1;
    case kI16: return 2;
    case kI32: return 4;
    case kF16: return 2;
    case kBF16: return 2;
    default: throw std::runtime_error("unknown data type id " + std::to_string(dtype));
  }
}

static float half_to_float(uint16_t h) {
  const uint32_t sign = (h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal
      exp = 127 - 15 + 1;
      while ((man & 0x400u) == 0) {
        man <<= 1;
        --exp;
      }
      man &= 0x3ffu;
      bits = sign | (exp << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

static float bf16_to_float(uint16_t h) {
  const uint32_t bits = static_cast<uint32_t>(h) << 16;
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

double Variable::as_number() const {
  if (data.empty()) throw std::runtime_error("variable " + name + " is empty");
  switch (dtype) {
    case kF32: { float v; std::memcpy(&v, data.data(), 4); return v; }
    case kI8: return static_cast<int8_t>(data[0]);
    case kI16: { int16_t v; std::memcpy(&v, data.data(), 2); return v; }
    case kI32: { int32_t v; std::memcpy(&v, data.data(), 4); return v; }
    case kF16: { uint16_t v; std::memcpy(&v, data.data(), 2); return half_to_float(v); }
    case kBF16: { uint16_t v; std::memcpy(&v, data.data(), 2); return bf16_to_float(v); }
  }
  return 0;
}

std::string Variable::as_string() const {
  return std::string(data.begin(), data.end());
}

std::vector<float> Variable::to_float() const {
  const int64_t n = size();
  std::vector<float> out(n);
  switch (dtype) {
    case kF32:
      std::memcpy(out.data(), data.data(), n * 4);
      break;
    case kI8:
      for (int64_t i = 0; i < n; ++i) out[i] = static_cast<int8_t>(data[i]);
      break;
    case kI16:
      for (int64_t i = 0; i < n; ++i) {
        int16_t v;
        std::memcpy(&v, data.data() + i * 2, 2);
        out[i] = v;
      }
      break;
    case kI32:
      for (int64_t i = 0; i < n; ++i) {
        int32_t v;
        std::memcpy(&v, data.data() + i * 4, 4);
        out[i] = static_cast<float>(v);
      }
      break;
    case kF16:
    case kBF16:
      for (int64_t i = 0; i < n; ++i) {
        uint16_t v;
        std::memcpy(&v, data.data() + i * 2, 2);
        out[i] = dtype == kF16 ? half_to_float(v) : bf16_to_float(v);
      }
      break;
  }
  return out;
}

// ---- JSON ---------------------------------------------------------------------------------

namespace {

struct JsonParser {
  const std::string& s;
  size_t p = 0;
  explicit JsonParser(const std::string& text) : s(text) {}

  [[noreturn]] void fail(const char* what) {
    throw std::runtime_error(std::string("JSON parse error: ") + what + " at offset " +
                             std::to_string(p));
  }
  void ws() {
    while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p;
  }
  static void append_utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) {
      out += static_cast<char>(cp);
    } else if (cp < 0x800) {
      out += static_cast<char>(0xC0 | (cp >> 6));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    } else if (cp < 0x10000) {
      out += static_cast<char>(0xE0 | (cp >> 12));
      out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    } else {
      out += static_cast<char>(0xF0 | (cp >> 18));
      out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
      out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    }
  }
  uint32_t hex4() {
    if (p + 4 > s.size()) fail("truncated \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = s[p++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad hex digit");
    }
    return v;
  }
  std::string string() {
    if (s[p] != '"') fail("expected string");
    ++p;
    std::string out;
    while (true) {
      if (p >= s.size()) fail("unterminated string");
      const char c = s[p++];
      if (c == '"') break;
      if (c != '\\') {
        out += c;
        continue;
      }
      if (p >= s.size()) fail("unterminated escape");
      const char e = s[p++];
      switch (e) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp <= 0xDBFF && p + 1 < s.size() && s[p] == '\\' && s[p + 1] == 'u') {
            p += 2;
            const uint32_t lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          append_utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
    return out;
  }
  Json value() {
    ws();
    if (p >= s.size()) fail("unexpected end");
    Json j;
    const char c = s[p];
    if (c == '{') {
      j.kind = Json::kObject;
      ++p;
      ws();
      if (p < s.size() && s[p] == '}') { ++p; return j; }
      while (true) {
        ws();
        std::string key = string();
        ws();
        if (p >= s.size() || s[p] != ':') fail("expected ':'");
        ++p;
        j.obj.emplace_back(std::move(key), value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == '}') { ++p; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      j.kind = Json::kArray;
      ++p;
      ws();
      if (p < s.size() && s[p] == ']') { ++p; return j; }
      while (true) {
        j.arr.emplace_back(value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == ']') { ++p; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      j.kind = Json::kString;
      j.str = string();
    } else if (s.compare(p, 4, "true") == 0) {
      j.kind = Json::kBool; j.b = true; p += 4;
    } else if (s.compare(p, 5, "false") == 0) {
      j.kind = Json::kBool; j.b = false; p += 5;
    } else if (s.compare(p, 4, "null") == 0) {
      j.kind = Json::kNull; p += 4;
    } else {
      size_t end = p;
      while (end < s.size() && (std::isdigit(static_cast<unsigned char>(s[end])) || s[end] == '-' ||
                                s[end] == '+' || s[end] == '.' || s[end] == 'e' || s[end] == 'E'))
        ++end;
      if (end == p) fail("unexpected character");
      j.kind = Json::kNumber;
      j.num = std::stod(s.substr(p, end - p));
      p = end;
    }
    return j;
  }
};

}  // namespace

Json Json::parse(const std::string& text) {
  JsonParser parser(text);
  Json j = parser.value();
  parser.ws();
  if (parser.p != text.size()) parser.fail("trailing characters");
  return j;
}

const Json* Json::get(const std::string& key) const {
  if (kind != kObject) return nullptr;
  for (const auto& kv : obj)
    if (kv.first == key) return &kv.second;
  return nullptr;
}

std::string read_text_file(const std::string& path, bool* found) {
  std::ifstream in(path, std::ios::binary);
  if (!in) {
    if (found) *found = false;
    return std::string();
  }
  if (found) *found = true;
  std::ostringstream ss;
  ss << in.rdbuf();
  return ss.str();
}

// ---- model.bin ------------------------------------------------------------------------------

namespace {

constexpr uint32_t kCurrentBinaryVersion = 6;  // include/ctranslate2/models/model.h

template <typename T>
T consume(std::istream& in) {
  T v;
  in.read(reinterpret_cast<char*>(&v), sizeof(T));
  if (!in) throw std::runtime_error("File model.bin is incomplete: failed to read a value");
  return v;
}

std::string consume_string(std::istream& in) {
  const uint16_t len = consume<uint16_t>(in);  // includes the trailing NUL
  std::string s(len, '\0');
  in.read(&s[0], len);
  if (!in) throw std::runtime_error("File model.bin is incomplete: failed to read a string");
  if (!s.empty() && s.back() == '\0') s.pop_back();
  return s;
}

int dtype_from_item_size(uint8_t item_size) {
  // Binary versions < 4 store the item size only (src/models/model.cc: get_dtype_from_item_size).
  switch (item_size) {
    case 4: return kF32;
    case 2: return kI16;
    case 1: return kI8;
    default:
      throw std::runtime_error("unknown data type of width " + std::to_string(item_size));
  }
}

}  // namespace

std::shared_ptr<ModelFile> ModelFile::load(const std::string& dir) {
  auto model = std::make_shared<ModelFile>();
  model->dir = dir;
  const std::string path = dir + "/model.bin";
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("Unable to open file 'model.bin' in model '" + dir + "'");

  model->binary_version = consume<uint32_t>(in);
  if (model->binary_version > kCurrentBinaryVersion)
    throw std::runtime_error("Unsupported model binary version. This executable supports models "
                             "with binary version v" + std::to_string(kCurrentBinaryVersion) +
                             " or below, but the model has binary version v" +
                             std::to_string(model->binary_version));
  if (model->binary_version >= 2) {
    model->spec = consume_string(in);
    model->revision = consume<uint32_t>(in);
  } else {
    model->revision = 1;
  }

  const uint32_t num_variables = consume<uint32_t>(in);
  model->variables.reserve(num_variables);
  for (uint32_t i = 0; i < num_variables; ++i) {
    Variable var;
    var.name = consume_string(in);
    const uint8_t rank = consume<uint8_t>(in);
    var.shape.resize(rank);
    for (uint8_t r = 0; r < rank; ++r) var.shape[r] = consume<uint32_t>(in);
    uint64_t num_bytes;
    if (model->binary_version >= 4) {
      var.dtype = consume<uint8_t>(in);
      num_bytes = consume<uint32_t>(in);
    } else {
      const uint8_t item_size = consume<uint8_t>(in);
      var.dtype = dtype_from_item_size(item_size);
      num_bytes = static_cast<uint64_t>(consume<uint32_t>(in)) * item_size;
    }
    if (num_bytes != static_cast<uint64_t>(var.size()) * dtype_size(var.dtype))
      throw std::runtime_error("Variable '" + var.name + "' has an invalid payload size");
    var.data.resize(num_bytes);
    in.read(var.data.data(), num_bytes);
    if (!in) throw std::runtime_error("File model.bin is incomplete: variable '" + var.name + "'");
    model->index_.emplace(var.name, model->variables.size());
    model->variables.emplace_back(std::move(var));
  }

  if (model->binary_version >= 3) {
    const uint32_t num_aliases = consume<uint32_t>(in);
    for (uint32_t i = 0; i < num_aliases; ++i) {
      std::string alias = consume_string(in);
      std::string target = consume_string(in);
      model->alias_map_.emplace(alias, target);
      model->aliases.emplace_back(std::move(alias), std::move(target));
    }
  }

  bool found = false;
  const std::string cfg = read_text_file(dir + "/config.json", &found);
  if (found) model->config = Json::parse(cfg);
  return model;
}

const Variable* ModelFile::find(const std::string& name) const {
  auto it = index_.find(name);
  if (it != index_.end()) return &variables[it->second];
  auto at = alias_map_.find(name);
  if (at != alias_map_.end()) {
    it = index_.find(at->second);
    if (it != index_.end()) return &variables[it->second];
  }
  return nullptr;
}

const Variable& ModelFile::get(const std::string& name) const {
  const Variable* v = find(name);
  if (!v) throw std::out_of_range("variable " + name + " not found");
  return *v;
}

double ModelFile::attribute(const std::string& name, double default_value) const {
  const Variable* v = find(name);
  if (!v) return default_value;
  return v->as_number();
}

bool ModelFile::flag(const std::string& name, bool default_value) const {
  return attribute(name, default_value ? 1 : 0) != 0;
}

std::string ModelFile::config_string(const std::string& key,
                                     const std::string& default_value) const {
  const Json* j = config.get(key);
  if (j && j->kind == Json::kString) return j->str;
  // binary_version < 6: tokens were attributes of the model (sequence_to_sequence.cc:60-64).
  const Variable* v = find(key);
  if (v && v->dtype == kI8) return v->as_string();
  return default_value;
}

bool ModelFile::config_bool(const std::string& key, bool default_value) const {
  const Json* j = config.get(key);
  if (j && j->kind == Json::kBool) return j->b;
  return default_value;
}

bool ModelFile::config_has_number(const std::string& key) const {
  const Json* j = config.get(key);
  return j && j->kind == Json::kNumber;
}

double ModelFile::config_number(const std::string& key, double default_value) const {
  const Json* j = config.get(key);
  if (j && j->kind == Json::kNumber) return j->num;
  return default_value;
}

// ---- vocabulary -----------------------------------------------------------------------------

std::shared_ptr<Vocabulary> Vocabulary::load(const std::string& dir, const std::string& stem,
                                             const std::string& unk, const std::string& bos,
                                             const std::string& eos) {
  auto vocab = std::make_shared<Vocabulary>();
  vocab->unk_ = unk;
  vocab->bos_ = bos;
  vocab->eos_ = eos;
  std::vector<std::string> tokens;
  bool found = false;
  std::string text = read_text_file(dir + "/" + stem + ".json", &found);
  if (found) {
    Json j = Json::parse(text);
    if (j.kind != Json::kArray) throw std::runtime_error(stem + ".json is not a JSON array");
    for (const auto& t : j.arr) tokens.push_back(t.str);
  } else {
    text = read_text_file(dir + "/" + stem + ".txt", &found);
    if (!found) return nullptr;
    // Vocabulary::from_text_file (src/vocabulary.cc:9-31): keep '\r' unless every line has one.
    bool remove_cr = true;
    size_t start = 0;
    while (start < text.size()) {
      size_t end = text.find('\n', start);
      if (end == std::string::npos) end = text.size();
      std::string line = text.substr(start, end - start);
      if (line.empty() || line.back() != '\r') remove_cr = false;
      tokens.emplace_back(std::move(line));
      start = end + 1;
    }
    if (remove_cr)
      for (auto& t : tokens) t.pop_back();
  }
  for (auto& t : tokens) {
    vocab->ids_.emplace(t, vocab->tokens_.size());  // first occurrence wins (emplace semantics)
    vocab->tokens_.emplace_back(std::move(t));
  }
  if (!vocab->contains(unk)) {
    vocab->ids_.emplace(unk, vocab->tokens_.size());
    vocab->tokens_.push_back(unk);
  }
  return vocab;
}

std::shared_ptr<Vocabulary> Vocabulary::from_tokens(std::vector<std::string> tokens, const std::string& unk,
                                                    const std::string& bos, const std::string& eos) {
  auto vocab = std::make_shared<Vocabulary>();
  vocab->unk_ = unk;
  vocab->bos_ = bos;
  vocab->eos_ = eos;
  for (auto& t : tokens) {
    vocab->ids_.emplace(t, vocab->tokens_.size());
    vocab->tokens_.emplace_back(std::move(t));
  }
  if (!vocab->contains(unk)) {
    vocab->ids_.emplace(unk, vocab->tokens_.size());
    vocab->tokens_.push_back(unk);
  }
  return vocab;
}

std::shared_ptr<ModelFile> ModelFile::from_memory(const std::string& spec, uint32_t revision,
                                                  uint32_t binary_version, const std::string& config_json,
                                                  std::vector<Variable> vars) {
  auto model = std::make_shared<ModelFile>();
  if (binary_version > kCurrentBinaryVersion)  // (the check of ModelFile::load)
    throw std::runtime_error("Unsupported model binary version. This executable supports models "
                             "with binary version v" + std::to_string(kCurrentBinaryVersion) +
                             " or below, but the model has binary version v" + std::to_string(binary_version));
  model->binary_version = binary_version;
  model->spec = spec;
  model->revision = revision == 0 ? 1 : revision;
  for (auto& v : vars) {
    if (static_cast<size_t>(v.size()) * dtype_size(v.dtype) != v.data.size())
      throw std::runtime_error("Variable " + v.name + " has an invalid payload size");
    model->index_.emplace(v.name, model->variables.size());  // (the first of two equal names wins)
    model->variables.emplace_back(std::move(v));
  }
  if (!config_json.empty()) model->config = Json::parse(config_json);
  return model;
}

const std::string& Vocabulary::to_token(size_t id) const {
  if (id >= tokens_.size())
    throw std::invalid_argument("Invalid token ID " + std::to_string(id) +
                                ": valid IDs are between 0 and " +
                                std::to_string(tokens_.size() - 1));
  return tokens_[id];
}

size_t Vocabulary::to_id(const std::string& token) const {
  auto it = ids_.find(token);
  if (it == ids_.end()) return ids_.at(unk_);
  return it->second;
}

}  // namespace ct2x
