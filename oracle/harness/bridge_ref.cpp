// bridge_ref.cpp — ABI transcript recorder.  OUR code; linked against the GENUINE reference bridge (cpp/bridge.cpp, cpp/janus.cpp and
// cpp/common compiled in place by oracle/Makefile) and oracle/_ref/libggml_ref.so.  Drives the nine cgo symbols of cpp/bridge.h:132-165
// through the scripted session of tests/golden/abi_script.json exactly as pkg/server/server.go would (config mode: initContext, then
// init), on the CPU path (gpu1..gpu4 = 0), and prints one JSON object per call with what the call returned.  Build container only.
//
// usage: bridge_ref <model.gguf> <script.txt>     script lines:  ctx <idx> <n_ctx> <n_predict> <hi> <lo>  |  init  |  infer <idx> <job> <hex prompt>
//                                                                | status <job> | count <job> | stop <idx> | seed <job> | evalms <job> | genms <job>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include "booster_bridge.h"     // the same nine prototypes as cpp/bridge.h:132-165

static std::string unhex(const std::string & h) { std::string s; for (size_t i = 0; i + 1 < h.size(); i += 2) s += (char) strtol(h.substr(i, 2).c_str(), nullptr, 16); return s; }
static std::string hex(const char * p) { std::string s; static const char * d = "0123456789abcdef"; for (; p && *p; ++p) { s += d[(unsigned char) *p >> 4]; s += d[(unsigned char) *p & 15]; } return s; }

int main(int argc, char ** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[2]); std::string line;
    std::map<int, void *> ctxs;
    static char empty[] = "";
    while (std::getline(f, line)) {
        std::istringstream is(line); std::string op; is >> op;
        if (op == "ctx") {
            int idx, n_ctx, n_predict; float hi, lo; is >> idx >> n_ctx >> n_predict >> hi >> lo;
            void * c = initContext(idx, argv[1], 4, 512, 0, 0, 0, 0, n_ctx, n_predict, 0, 0.0f, 0.0f, 0.8f, 40, 0.9f, 1.0f, 1.1f, 64, 1, 200, 0.97f, hi, lo, 42, empty);
            ctxs[idx] = c;
            printf("{\"op\":\"ctx\",\"idx\":%d,\"ok\":%d}\n", idx, c != nullptr);
        } else if (op == "init") { init(empty, empty); printf("{\"op\":\"init\"}\n"); }
        else if (op == "infer") {
            int idx; std::string job, hx; is >> idx >> job >> hx;
            std::string prompt = unhex(hx);
            char sess[] = "sess";
            const long long n = doInference(idx, ctxs[idx], (char *) job.c_str(), sess, (char *) prompt.c_str());
            printf("{\"op\":\"infer\",\"job\":\"%s\",\"ret\":%lld}\n", job.c_str(), n);
        } else if (op == "status") { std::string job; is >> job; printf("{\"op\":\"status\",\"job\":\"%s\",\"hex\":\"%s\"}\n", job.c_str(), hex(status((char *) job.c_str())).c_str()); }
        else if (op == "count") { std::string job; is >> job; printf("{\"op\":\"count\",\"job\":\"%s\",\"ret\":%lld}\n", job.c_str(), (long long) getPromptTokenCount((char *) job.c_str())); }
        else if (op == "stop") { int idx; is >> idx; stopInference(idx); printf("{\"op\":\"stop\",\"idx\":%d}\n", idx); }
        else if (op == "seed") { std::string job; is >> job; printf("{\"op\":\"seed\",\"job\":\"%s\",\"nonzero\":%d}\n", job.c_str(), getSeed((char *) job.c_str()) != 0); }
        else if (op == "evalms") { std::string job; is >> job; printf("{\"op\":\"evalms\",\"job\":\"%s\",\"nonneg\":%d}\n", job.c_str(), promptEval((char *) job.c_str()) >= 0); }
        else if (op == "genms") { std::string job; is >> job; printf("{\"op\":\"genms\",\"job\":\"%s\",\"nonneg\":%d}\n", job.c_str(), timing((char *) job.c_str()) >= 0); }
        fflush(stdout);
    }
    return 0;
}
