"""TaskList / TaskID / Task: the operator surface the hot path plugs into.

Mirror of src/tasklist/task_list.hpp:30-236: tasks are callables `f(driver, stage) ->
TaskStatus`, registered with a dependency bitset, executed by DoAvailable() until all are
complete (Driver::ExecuteTaskList, src/driver/driver.cpp:290-307).
"""
import enum

NUMBER_TASKID_BITS = 64


class TaskStatus(enum.Enum):
    fail = 0
    complete = 1
    incomplete = 2


class TaskListStatus(enum.Enum):
    running = 0
    stuck = 1
    complete = 2
    nothing_to_do = 3


class TaskID:
    """64-bit bitset task identifier (task_list.hpp:37-81)."""
    __slots__ = ("bits",)

    def __init__(self, id=0):
        if id == 0:
            self.bits = 0
        else:
            if id > NUMBER_TASKID_BITS:
                raise ValueError("more than %d tasks in a TaskList" % NUMBER_TASKID_BITS)
            self.bits = 1 << (id - 1)

    def Clear(self):
        self.bits = 0

    def CheckDependencies(self, dep):
        return (self.bits & dep.bits) == dep.bits

    def SetComplete(self, rhs):
        self.bits |= rhs.bits

    def __eq__(self, rhs):
        return self.bits == rhs.bits

    def __ne__(self, rhs):
        return self.bits != rhs.bits

    def __or__(self, rhs):
        r = TaskID()
        r.bits = self.bits | rhs.bits
        return r

    def __xor__(self, rhs):
        r = TaskID()
        r.bits = self.bits ^ rhs.bits
        return r

    def __and__(self, rhs):
        r = TaskID()
        r.bits = self.bits & rhs.bits
        return r

    def __hash__(self):
        return hash(self.bits)

    def __repr__(self):
        return "TaskID(%s)" % format(self.bits, "064b")


class Task:
    """task_list.hpp:88-110"""

    def __init__(self, id, dep, func, name=""):
        self.myid_, self.dep_, self.func_, self.name = id, dep, func, name
        self.complete_ = False

    def __call__(self, d, s):
        return self.func_(d, s)

    def GetID(self):
        return self.myid_

    def GetDependency(self):
        return self.dep_

    def SetComplete(self):
        self.complete_ = True

    def SetIncomplete(self):
        self.complete_ = False

    def IsComplete(self):
        return self.complete_

    def ChangeDependency(self, id, newdep):
        if (self.dep_ & id) == id:
            self.dep_ = (self.dep_ ^ id) | newdep


class TaskList:
    """task_list.hpp:117-236"""

    def __init__(self):
        self.task_list_ = []
        self.tasks_completed_ = TaskID()

    def IsComplete(self):
        return all(self.tasks_completed_.CheckDependencies(t.GetID()) for t in self.task_list_)

    def Size(self):
        return len(self.task_list_)

    def Empty(self):
        return not self.task_list_

    def MarkTaskComplete(self, id):
        self.tasks_completed_.SetComplete(id)

    def GetIDLastTask(self):
        return self.task_list_[-1].GetID()

    def Reset(self):
        self.tasks_completed_.Clear()
        for t in self.task_list_:
            t.SetIncomplete()

    def DoAvailable(self, d, s):
        for task in self.task_list_:
            if self.tasks_completed_.CheckDependencies(task.GetDependency()) and not task.IsComplete():
                status = task(d, s)
                if status == TaskStatus.fail:
                    raise RuntimeError("### FATAL ERROR task '%s' failed" % task.name)
                if status == TaskStatus.complete:
                    task.SetComplete()
                    self.MarkTaskComplete(task.GetID())
        if self.IsComplete():
            return TaskListStatus.complete
        return TaskListStatus.running

    def AddTask(self, func, dep, name=None):
        """tl.AddTask(obj.Method, dependency) -> TaskID (member-function form :178-185)."""
        id = TaskID(len(self.task_list_) + 1)
        self.task_list_.append(Task(id, dep, func, name or getattr(func, "__name__", "task")))
        return id

    def InsertTask(self, func, dep, loc, name=None):
        """Insert BEFORE the task with ID loc and re-point dependencies (:212-234)."""
        for idx, t in enumerate(self.task_list_):
            if t.GetID() == loc:
                id = TaskID(len(self.task_list_) + 1)
                old_dep = t.GetDependency()
                self.task_list_.insert(idx, Task(id, dep, func, name or getattr(func, "__name__", "task")))
                for t2 in self.task_list_:
                    if t2.GetID() != id:
                        t2.ChangeDependency(old_dep, id)
                return id
        return TaskID(0)
