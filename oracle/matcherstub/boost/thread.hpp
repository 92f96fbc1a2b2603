// oracle/matcherstub: <boost/thread.hpp> as include/MapPoint.h and src/MapPoint.cc use it — boost::mutex with scoped_lock; single-threaded
// test infrastructure, so locking is a no-op (TEST INFRASTRUCTURE ONLY).
#ifndef ORB_ORACLE_MATCHERSTUB_BOOST_THREAD_HPP
#define ORB_ORACLE_MATCHERSTUB_BOOST_THREAD_HPP
namespace boost {
class mutex {
public:
    class scoped_lock {
    public:
        explicit scoped_lock(mutex&) {}
    };
};
}
#endif
