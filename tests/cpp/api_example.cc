// Test program of the C++ host API (include/ct2_mi355.hpp): the reference's own usage example
// (README "C++" section / tests/translator_test.cc:53-97) with the namespace swapped.
//   api_example translate <model_dir> <beam> <token> ...      -> hypothesis tokens + score
//   api_example generate  <model_dir> <max_length> <token> ... -> generated tokens
//   api_example score     <model_dir> <n_src> <src tokens> <tgt tokens> -> token log-probabilities
//   api_example whisper   <model_dir> <n_mels> <frames>        -> models::Whisper generate / detect_language
//   api_example reject    <model_dir>                          -> an unsupported option must throw
// Exit code 0 on success, 2 when the backend reports an error (printed on stderr).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "ct2_mi355.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: api_example translate|generate|reject <model_dir> ...\n");
    return 1;
  }
  const std::string mode = argv[1], model = argv[2];
  try {
    if (mode == "translate") {
      ct2_mi355::Translator translator(model, ct2_mi355::Device::CUDA, ct2_mi355::ComputeType::DEFAULT, {0});
      ct2_mi355::TranslationOptions options;
      options.beam_size = static_cast<size_t>(std::atoi(argv[3]));
      options.return_scores = true;
      const std::vector<std::vector<std::string>> batch = {std::vector<std::string>(argv + 4, argv + argc)};
      const auto results = translator.translate_batch(batch, options);
      for (const auto& token : results[0].output()) std::cout << token << ' ';
      std::cout << "| " << results[0].score() << " | " << translator.compute_type() << std::endl;
      // target prefix: the first two tokens of the hypothesis must come back as its prefix
      const auto& hyp = results[0].output();
      if (hyp.size() >= 2) {
        const std::vector<std::vector<std::string>> prefix = {{hyp[0], hyp[1]}};
        const auto again = translator.translate_batch(batch, prefix, options);
        if (again[0].output() != hyp) {
          std::fprintf(stderr, "target_prefix changed the hypothesis\n");
          return 3;
        }
      }
    } else if (mode == "generate") {
      ct2_mi355::Generator generator(model, ct2_mi355::Device::CUDA);
      ct2_mi355::GenerationOptions options;
      options.max_length = static_cast<size_t>(std::atoi(argv[3]));
      options.include_prompt_in_result = false;
      const std::vector<std::vector<std::string>> batch = {std::vector<std::string>(argv + 4, argv + argc)};
      const auto results = generator.generate_batch(batch, options);
      for (const auto& token : results[0].sequences[0]) std::cout << token << ' ';
      std::cout << std::endl;
    } else if (mode == "generate_prompt") {
      // api_example generate_prompt <model_dir> <max_length> <prompt tokens...>: the reference's
      // default GenerationOptions (include_prompt_in_result = true) + return_scores
      ct2_mi355::Generator generator(model, ct2_mi355::Device::CUDA);
      ct2_mi355::GenerationOptions options;
      options.max_length = static_cast<size_t>(std::atoi(argv[3]));
      options.return_scores = true;
      const std::vector<std::vector<std::string>> batch = {std::vector<std::string>(argv + 4, argv + argc)};
      const auto results = generator.generate_batch(batch, options);
      for (const auto& token : results[0].sequences[0]) std::cout << token << ' ';
      std::cout << "| " << results[0].scores[0] << std::endl;
    } else if (mode == "alternatives") {
      // api_example alternatives <model_dir> <max_length> <num_hypotheses> <prompt tokens...>:
      // GenerationOptions::return_alternatives with the prompt in the result (the reference's default)
      ct2_mi355::Generator generator(model, ct2_mi355::Device::CUDA);
      ct2_mi355::GenerationOptions options;
      options.max_length = static_cast<size_t>(std::atoi(argv[3]));
      options.num_hypotheses = static_cast<size_t>(std::atoi(argv[4]));
      options.return_alternatives = true;
      options.return_scores = true;
      const std::vector<std::vector<std::string>> batch = {std::vector<std::string>(argv + 5, argv + argc)};
      const auto results = generator.generate_batch(batch, options);
      for (size_t k = 0; k < results[0].sequences.size(); ++k) {
        for (const auto& token : results[0].sequences[k]) std::cout << token << ' ';
        std::cout << "| " << results[0].scores[k] << std::endl;
      }
    } else if (mode == "score") {
      // api_example score <model_dir> <n_source> <source tokens...> <target tokens...>
      ct2_mi355::Translator translator(model);
      const int n_src = std::atoi(argv[3]);
      const std::vector<std::vector<std::string>> source = {std::vector<std::string>(argv + 4, argv + 4 + n_src)};
      const std::vector<std::vector<std::string>> target = {std::vector<std::string>(argv + 4 + n_src, argv + argc)};
      const auto results = translator.score_batch(source, target);
      for (size_t i = 0; i < results[0].tokens.size(); ++i)
        std::cout << results[0].tokens[i] << ' ' << results[0].tokens_score[i] << std::endl;
    } else if (mode == "whisper") {
      // api_example whisper <model_dir> <n_mels> <frames>: greedy transcription of a deterministic
      // feature pattern with the prompt <|startoftranscript|> <|en|> <|transcribe|> <|notimestamps|>
      ct2_mi355::models::Whisper whisper(model, ct2_mi355::Device::CUDA, ct2_mi355::ComputeType::FLOAT32);
      const size_t n_mels = static_cast<size_t>(std::atoi(argv[3])), frames = static_cast<size_t>(std::atoi(argv[4]));
      std::vector<float> data(2 * n_mels * frames);
      for (size_t i = 0; i < data.size(); ++i)
        data[i] = static_cast<float>(static_cast<int>((i * 37 + 11) % 101) - 50) / 40.f;
      ct2_mi355::models::WhisperFeatures f{data.data(), 2, n_mels, frames};
      ct2_mi355::models::WhisperOptions options;
      options.beam_size = 1;
      options.max_length = 24;
      options.return_scores = true;
      options.return_no_speech_prob = true;
      const std::vector<std::vector<std::string>> prompts(
          2, {"<|startoftranscript|>", "<|en|>", "<|transcribe|>", "<|notimestamps|>"});
      const auto results = whisper.generate(f, prompts, options);
      for (const auto& r : results) {
        for (size_t id : r.sequences_ids[0]) std::cout << id << ' ';
        std::cout << "| " << r.scores[0] << " | " << r.no_speech_prob << std::endl;
      }
      const auto langs = whisper.detect_language(f);
      std::cout << langs[0][0].first << ' ' << langs[0][0].second << std::endl;
      std::cout << whisper.n_mels() << ' ' << whisper.num_languages() << ' ' << whisper.is_multilingual() << std::endl;
    } else if (mode == "reject") {
      ct2_mi355::Translator translator(model);
      ct2_mi355::TranslationOptions options;
      try {  // one source, two prefixes
        translator.translate_batch({{"a"}}, {{"b"}, {"c"}}, options);
      } catch (const std::invalid_argument& e) {
        std::cout << "rejected: " << e.what() << std::endl;
        return 0;
      }
      return 4;
    } else if (mode == "callback") {
      // TranslationOptions::callback: every generated token is handed over in step order; the
      // second sentence is stopped after its 2nd token; argv[3] = tokens of the first source
      // sentence, the rest is split evenly over two sentences
      ct2_mi355::Translator translator(model);
      ct2_mi355::TranslationOptions options;
      options.beam_size = 1;
      options.return_scores = true;
      options.callback = [](ct2_mi355::GenerationStepResult r) {
        std::cout << r.batch_id << ' ' << r.step << ' ' << r.token << ' ' << r.token_id << ' '
                  << (r.score ? *r.score : 1.f) << ' ' << r.is_last << std::endl;
        return r.batch_id == 1 && r.step == 1;
      };
      const int n0 = std::atoi(argv[3]);
      std::vector<std::string> a, b;
      for (int i = 4; i < argc; ++i) (i - 4 < n0 ? a : b).push_back(argv[i]);
      const auto results = translator.translate_batch({a, b}, options);
      for (const auto& r : results) {
        std::cout << "=";
        for (const auto& tok : r.hypotheses[0]) std::cout << ' ' << tok;
        std::cout << std::endl;
      }
      {  // return_logits_vocab: the arg-max of every step's logits is the step's token
        ct2_mi355::TranslationOptions lo;
        lo.beam_size = 1;
        lo.return_logits_vocab = true;
        lo.return_end_token = true;
        const auto r = translator.translate_batch({a}, lo);
        std::cout << "logits:";
        for (const auto& step : r[0].logits[0]) {
          size_t best = 0;
          for (size_t j = 1; j < step.size(); ++j)
            if (step[j] > step[best]) best = j;
          std::cout << ' ' << best;
        }
        std::cout << " |";
        for (const auto& tok : r[0].hypotheses[0]) std::cout << ' ' << tok;
        std::cout << std::endl;
      }
      options.beam_size = 2;
      try {
        translator.translate_batch({a}, options);
      } catch (const std::exception& e) {
        std::cout << "rejected: " << e.what() << std::endl;
      }
      options.beam_size = 1;
      options.callback = [](ct2_mi355::GenerationStepResult) -> bool { throw std::logic_error("from the callback"); };
      try {
        translator.translate_batch({a}, options);
      } catch (const std::logic_error& e) {
        std::cout << "raised: " << e.what() << std::endl;
      }
    } else if (mode == "replace_unknowns") {
      // TranslationOptions::replace_unknowns: the prefix forces an unknown target token, which
      // comes back as the source token under the attention's maximum; argv[3..] = source tokens
      ct2_mi355::Translator translator(model);
      ct2_mi355::TranslationOptions options;
      options.beam_size = 1;
      options.replace_unknowns = true;
      std::vector<std::string> source;
      for (int i = 3; i < argc; ++i) source.push_back(argv[i]);
      const auto results = translator.translate_batch({source}, {{"<unk>"}}, options);
      for (const auto& tok : results[0].hypotheses[0]) std::cout << tok << ' ';
      std::cout << "| " << results[0].attention.size() << std::endl;
    } else {
      return 1;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
  return 0;
}
