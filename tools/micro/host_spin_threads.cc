#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main(int argc,char**argv){int NT=atoi(argv[1]); int mode=atoi(argv[2]); auto T0=std::chrono::steady_clock::now();
 auto work=[&](){ if(mode==0){volatile double x=1; for(long i=0;i<300000000;i++) x=x*1.0000001;} else { for(int r=0;r<2000000;r++){ std::vector<int> v; v.push_back(r); std::vector<int> w; w.push_back(r);} } };
 std::vector<std::thread> p; for(int t=1;t<NT;t++)p.emplace_back(work); work(); for(auto&t:p)t.join();
 printf("threads %d mode %d wall %.1f ms\n",NT,mode,1e3*std::chrono::duration<double>(std::chrono::steady_clock::now()-T0).count());}
