// Host-side model container: reads a converted CTranslate2 model directory unchanged
// (model.bin + config.json + vocabularies), exactly the wire format Model::load consumes
// (src/models/model.cc:594-661,767-777; writer python/ctranslate2/specs/model_spec.py:382-415).
// No device code here: this file is also exercised by the CPU-only test-suite.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace ct2x {

// DataType ids as stored in model.bin (include/ctranslate2/types.h:16-23).
enum DType { kF32 = 0, kI8 = 1, kI16 = 2, kI32 = 3, kF16 = 4, kBF16 = 5 };

size_t dtype_size(int dtype);

struct Variable {
  std::string name;
  std::vector<int64_t> shape;
  int dtype = kF32;
  std::vector<char> data;

  int64_t size() const {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    return n;
  }
  bool is_scalar() const { return shape.empty() || size() == 1; }
  // Scalar attribute helpers (attributes are stored as 0-d int8/int16/int32/float variables).
  double as_number() const;
  std::string as_string() const;  // int8 byte strings (legacy vocabulary tokens)
  // Returns the content converted to float32 (f32/f16/bf16/int8/int16/int32 sources).
  std::vector<float> to_float() const;
};

// Minimal JSON value (config.json / *_vocabulary.json are flat objects / string arrays).
struct Json {
  enum Kind { kNull, kBool, kNumber, kString, kArray, kObject } kind = kNull;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* get(const std::string& key) const;
  static Json parse(const std::string& text);  // throws std::runtime_error on bad input
};

class ModelFile {
public:
  // Throws std::runtime_error with the reference's wording where practical
  // ("Unsupported model binary version", "has an invalid payload size", ...).
  static std::shared_ptr<ModelFile> load(const std::string& dir);
  // The same container from variables that are already in memory (the reference's
  // ModelMemoryReader use case, include/ctranslate2/models/model_reader.h:36-55; also how a host
  // application that has loaded the model itself hands it to this backend): `vars` are copied,
  // `config_json` may be empty. dir stays empty: vocabularies are given separately.
  static std::shared_ptr<ModelFile> from_memory(const std::string& spec, uint32_t revision,
                                                uint32_t binary_version, const std::string& config_json,
                                                std::vector<Variable> vars);

  std::string dir;
  uint32_t binary_version = 0;
  std::string spec;       // "TransformerSpec", "TransformerBase", "TransformerDecoderSpec", ...
  uint32_t revision = 0;
  std::vector<Variable> variables;                       // in file order
  std::vector<std::pair<std::string, std::string>> aliases;  // (alias, target) in file order
  Json config;                                           // config.json (kNull when absent)

  const Variable* find(const std::string& name) const;   // resolves aliases
  const Variable& get(const std::string& name) const;    // throws if missing
  bool has(const std::string& name) const { return find(name) != nullptr; }

  // Model::get_attribute_with_default / get_flag_with_default (src/models/model.cc:237-272).
  double attribute(const std::string& name, double default_value) const;
  bool flag(const std::string& name, bool default_value) const;

  // config.json accessors with the binary_version < 6 fallbacks of
  // SequenceToSequenceModel::initialize (src/models/sequence_to_sequence.cc:60-77).
  std::string config_string(const std::string& key, const std::string& default_value) const;
  bool config_bool(const std::string& key, bool default_value) const;
  bool config_has_number(const std::string& key) const;
  double config_number(const std::string& key, double default_value) const;

  // include/ctranslate2/models/model.h:87-89: activation quantisation rounds to nearest only
  // for binary_version >= 5; older models truncate.
  bool round_before_cast_in_quantization() const { return binary_version >= 5; }

private:
  std::unordered_map<std::string, size_t> index_;
  std::unordered_map<std::string, std::string> alias_map_;
};

// Vocabulary (include/ctranslate2/vocabulary.h:16-70, src/vocabulary.cc).
class Vocabulary {
public:
  static std::shared_ptr<Vocabulary> load(const std::string& dir, const std::string& stem,
                                          const std::string& unk, const std::string& bos,
                                          const std::string& eos);  // nullptr if absent
  static std::shared_ptr<Vocabulary> from_tokens(std::vector<std::string> tokens, const std::string& unk,
                                                 const std::string& bos, const std::string& eos);
  size_t size() const { return tokens_.size(); }
  const std::string& to_token(size_t id) const;
  size_t to_id(const std::string& token) const;  // unknown -> unk id
  bool contains(const std::string& token) const { return ids_.count(token) != 0; }
  size_t unk_id() const { return to_id(unk_); }
  size_t bos_id() const { return to_id(bos_); }
  size_t eos_id() const { return to_id(eos_); }

private:
  std::vector<std::string> tokens_;
  std::unordered_map<std::string, size_t> ids_;
  std::string unk_, bos_, eos_;
};

std::string read_text_file(const std::string& path, bool* found);

}  // namespace ct2x
