// mvolps -- command-line front end over libmvolps_amd.so; counterpart of MVOLPS's main
// (/root/reference/2test.cpp:13-157): same flags, same dispatch on the file extension
// (2test.cpp:65-81), then initProblem (util.cpp:277-292) and branchAndBound (bs.cpp:54).
// Extra flags of this build: --repaired (reference_quirks = 0), --max-nodes N, --events FILE.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/mvx_bnb.h"

namespace {
class InputParser { // InputParser.h:8-35
public:
  InputParser(int argc, char **argv) {
    for (int i = 1; i < argc; i++) tokens.push_back(std::string(argv[i]));
  }
  const std::string &getCMDOption(const std::string &option) const {
    auto itr = std::find(tokens.begin(), tokens.end(), option);
    if (itr != tokens.end() && ++itr != tokens.end()) return *itr;
    static const std::string empty("");
    return empty;
  }
  bool CMDOptionExists(const std::string &option) const { return std::find(tokens.begin(), tokens.end(), option) != tokens.end(); }

private:
  std::vector<std::string> tokens;
};
} // namespace

int main(int argc, char **argv) {
  InputParser input(argc, argv);
  if (input.CMDOptionExists("-h") || input.CMDOptionExists("--help")) {
    std::cout << "Usage: mvolps [OPTION]\n"
              << "File input options:\n"
              << "  -f/--file [FILENAME.{mps|lp}]\n\n"
              << "  --events [FILE] write the B&B event stream (one line per event) to FILE\n"
              << "Output verbosity options:\n"
              << "  -s/--silent\n  -v/--verbose\n  -d/--debug\n  -so/--solver-output\n\n"
              << "Algorithm strategy options:\n"
              << "  -vs [{0|1|2}]\n"
              << "    0. vars are picked on order\n"
              << "    1. vars are picked on fractional part closeness to 0.5\n"
              << "    2. vars are picked on greatest impact on obj. function\n"
              << "  -bs [{0|1}]\n"
              << "    0. nodes are picked for DFS (FIFO/queue)\n"
              << "    1. nodes are picked for best-FS (greatest z-value)\n"
              << "  -cm [{0|1}]\n"
              << "    0. disable cut generation\n"
              << "    1. generate Gomory mixed integer\n"
              << "    -cf [0...1]\n"
              << "      Percentage of generated cuts to be added per node\n"
              << "  --repaired      children keep the opposite bound, integrality within 1e-9, repaired GMI cuts\n"
              << "  --cut-select K  with --repaired -cm 1: 0 add the last cut, 1 add the -cf fraction of most effective cuts\n"
              << "  --max-nodes N   stop after N loop iterations\n"
              << "Help:\n  -h/--help\n";
    return 0;
  }
  mvx_term_out(MVX_OFF); // 2test.cpp:45
  const bool verbose = input.CMDOptionExists("-v") || input.CMDOptionExists("--verbose") || input.CMDOptionExists("-d") ||
                       input.CMDOptionExists("--debug");
  if (input.CMDOptionExists("-d") || input.CMDOptionExists("--debug") || input.CMDOptionExists("-so") ||
      input.CMDOptionExists("--solver-output"))
    mvx_term_out(MVX_ON);

  if (!(input.CMDOptionExists("-f") || input.CMDOptionExists("--file"))) {
    std::cout << "see ./mvolps -h for usage\n";
    return 0;
  }
  std::string fn = input.getCMDOption("-f");
  if (fn.empty()) fn = input.getCMDOption("--file");
  const size_t dot = fn.rfind('.');
  const std::string ext = dot == std::string::npos ? "" : fn.substr(dot);
  mvx_prob *prob = mvx_create_prob();
  if (ext == ".lp") {
    if (mvx_read_lp(prob, nullptr, fn.c_str())) std::exit(-1); // util.cpp:284-287
  } else if (ext == ".mps") {
    if (mvx_read_mps(prob, 2 /*GLP_MPS_FILE*/, nullptr, fn.c_str())) std::exit(-1); // util.cpp:290-292
  } else {
    std::cout << "Unrecognized filetype\n";
    return -1;
  }
  if (verbose) std::printf("%s\nProblem contains %d integer variables\n", mvx_version(), mvx_get_num_int(prob)); // util.cpp:278,298

  mvx_bnb_params params;
  mvx_bnb_default_params(&params);
  auto int_opt = [&](const char *flag, int lo, int hi, int *out) -> bool {
    if (!input.CMDOptionExists(flag)) return true;
    const int v = std::atoi(input.getCMDOption(flag).c_str());
    if (v < lo || v > hi) {
      std::fprintf(stderr, "Unknown parameter value for %s\n", flag);
      return false;
    }
    *out = v;
    return true;
  };
  if (!int_opt("-bs", 0, 1, &params.node_strat)) return -1; // 2test.cpp:92-104
  if (!int_opt("-vs", 0, 2, &params.var_strat)) return -1;  // 2test.cpp:106-121
  if (!int_opt("-cm", 0, 1, &params.cut_strat)) return -1;  // 2test.cpp:123-134
  if (input.CMDOptionExists("-cm")) {
    params.cut_chance = 1.0;
    if (input.CMDOptionExists("-cf")) {
      const double chance = std::atof(input.getCMDOption("-cf").c_str());
      if (!((chance <= 1.0) && (chance >= 0.0))) {
        std::fprintf(stderr, "Cut Frequency parameter must be in range [0.0, 1.0]\n");
        return -1;
      }
      params.cut_chance = chance; // stored, never read (util.cpp:259-261)
    }
  }
  if (input.CMDOptionExists("--repaired")) params.reference_quirks = 0;
  if (input.CMDOptionExists("--cut-select")) params.cut_select = std::atoi(input.getCMDOption("--cut-select").c_str());
  if (input.CMDOptionExists("--window")) params.window = std::atoi(input.getCMDOption("--window").c_str());
  if (input.CMDOptionExists("--max-nodes")) params.max_nodes = std::atoi(input.getCMDOption("--max-nodes").c_str());
  if (input.CMDOptionExists("--server"))
    std::fprintf(stderr, "--server: the ZeroMQ sink is not part of this build; use --events FILE for the same stream\n");

  mvx_bnb_result res;
  mvx_branchAndBound(nullptr, prob, &params, &res);
  if (input.CMDOptionExists("--events")) mvx_bnb_write_events(&res, input.getCMDOption("--events").c_str());
  mvx_bnb_print_tree(&res, nullptr); // bs.cpp:329-343
  std::vector<char> buf(64 + 64 * (size_t)res.n);
  mvx_bnb_solution_string(nullptr, prob, &res, buf.data(), (int)buf.size());
  std::printf("\n%s\n", buf.data()); // bs.cpp:345
  if (verbose) std::printf("Solution found after %d iterations (%lld pivots)\n", res.count, res.total_pivots);
  const int limit = res.hit_limit;
  mvx_bnb_free_result(&res);
  mvx_delete_prob(prob);
  return limit ? -1 : 0;
}
